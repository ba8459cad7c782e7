"""K6 transform kernels alone at the episode's shapes: time, algorithmic bytes (1R + 2.25W or 2.25R + 1W
of the activation tensor), share of the 8 TB/s HBM peak.   python tools/bench_wino_transforms.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import winograd as wg  # noqa: E402

dev = torch.device("cuda:0")


def gpu_time(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


tot = 0.0
for (C, H) in ((128, 112), (256, 56), (512, 28)):
    x = torch.randn(37, C, H, H, device=dev)
    V = wg._input(4, x)
    nbytes = 3.25 * x.numel() * 4
    for name, fn in (("input", lambda: wg._input(4, x)), ("output", lambda: wg._output(4, V, 37, H, H)),
                     ("grad-output", lambda: wg._grad_output(4, x))):
        t = gpu_time(fn)
        tot += t
        print(f"K6 {name:12s} (37,{C},{H},{H}) {t * 1e6:8.1f} us {nbytes / t / 1e9:7.0f} GB/s {100 * nbytes / t / 8e12:5.1f} %")
print(f"sum {tot * 1e6:.1f} us")
