"""The per-kernel rooflines of bench.py (`kernels` of its JSON line) as a table, without the episode workloads.
Usage (GPU box): python tools/kernel_table.py [substring ...] > profiles/r03/kernel_table_<what>.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    want = sys.argv[1:]
    from fpsg_amd import gemm_tuning
    gemm_tuning.enable()
    rows = bench.kernel_rooflines(torch.device("cuda:0"))
    print(f"{'kernel':44s} {'bound':5s} {'achieved':>10s} {'peak':>8s} {'unit':8s} {'frac':>6s} {'us':>9s}  shape")
    for name, e in rows.items():
        if want and not any(w in name for w in want):
            continue
        if not isinstance(e, dict) or "achieved" not in e:
            continue
        print(f"{name:44s} {e['bound']:5s} {e['achieved']:10.1f} {e['peak']:8.1f} {e['unit']:8s} {e['frac']:6.3f} "
              f"{e['us']:9.1f}  {e.get('shape', '')}")


if __name__ == "__main__":
    main()
