"""The evaluation loop's leg of bench.py (``configs.eval``) on its own: items/s of ``_return_reconstruction`` on the
configs[2] episode.  Usage (GPU box): python tools/eval_loop.py [items]; under rocprofv3 see tools/prof_eval.sh."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fpsg_amd import gemm_tuning, metrics  # noqa: E402


def main():
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    gemm_tuning.enable()
    probe = bench.EventProbe()
    metrics.set_launch_probe(probe)
    print(json.dumps(bench.eval_leg(torch.device("cuda:0"), probe, items=items)))


if __name__ == "__main__":
    main()
