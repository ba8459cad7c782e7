"""A/B of VERDICT r4 item 4: BatchNorm's forward finalize in the last-arriving workgroup of the statistics kernel
(``FPSG_BN_FINALIZE_FOLD=1``) against the two-launch form, through ``fpsg_bn_act_fwd`` at shapes of the c5 step that take the sliced
path (more than 16,384 values per channel).  Eager calls back to back and the same calls replayed as a hipGraph (how the
step runs them).

    python tools/bench_bn_fold.py
"""
from __future__ import annotations

import json
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd.fused_bn import bn_act  # noqa: E402


def _time(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    shapes = [(37, 64, 112, 112), (37, 128, 56, 56), (37, 256, 28, 28), (16, 769, 4736), (37, 128, 2048), (64, 1024, 2048)]
    for shape in shapes:
        C = shape[1]
        bn = (nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d)(C).to(dev).train()
        x = torch.randn(*shape, device=dev)
        row = {"shape": list(shape)}
        for fold in ("0", "1"):
            os.environ["FPSG_BN_FINALIZE_FOLD"] = fold
            with torch.no_grad():
                eager = sorted(_time(lambda: bn_act(bn, x, "relu"), 50) for _ in range(5))[2]
                # ten calls in one graph: the gap between the launches of a replayed graph is what the fold removes
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    for _ in range(3):
                        bn_act(bn, x, "relu")
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(10):
                        y = bn_act(bn, x, "relu")
                graph = sorted(_time(g.replay, 20) / 10 for _ in range(5))[2]
            row["fold" if fold == "1" else "two_launch"] = {"eager_us": round(eager, 2), "graph_us": round(graph, 2)}
        row["gain_graph_us"] = round(row["two_launch"]["graph_us"] - row["fold"]["graph_us"], 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
