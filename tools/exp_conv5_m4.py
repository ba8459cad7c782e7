"""Experiment: the 14x14 stage (conv5_x, 512 -> 512, 37 images) as F(4x4,3x3) on a 16x16 padded tile grid (16 tiles per image)
instead of F(2x2,3x3) (49 tiles per image): products and transforms (16x16 images as a proxy for the ragged 14x14 case)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import gemm_tuning  # noqa: E402
from fpsg_amd import winograd as wg  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


gemm_tuning.enable(path="gpurun_out/exp_conv5_records.csv", tune=True)
n = 37
for name, a2, P in (("m=2 (49 tiles)", 16, n * 49), ("m=4 (16 tiles)", 36, n * 16)):
    U = torch.randn(a2, 512, 512, device=dev)
    V = torch.randn(a2, 512, P, device=dev)
    gM = torch.randn(a2, 512, P, device=dev)
    tf = t(lambda: torch.bmm(U, V))
    tw = t(lambda: torch.bmm(gM, V.transpose(1, 2)))
    print(f"{name}: product fwd/dgrad {tf:7.1f} us, weight-gradient product {tw:7.1f} us", flush=True)
x14 = torch.randn(n, 512, 14, 14, device=dev)
x16 = torch.randn(n, 512, 16, 16, device=dev)
for name, m, x in (("m=2 @14", 2, x14), ("m=4 @16 (proxy)", 4, x16)):
    V = wg._input(m, x)
    ti = t(lambda: wg._input(m, x))
    to = t(lambda: wg._output(m, V, n, x.shape[2], x.shape[3]))
    tg = t(lambda: wg._grad_output(m, x))
    print(f"{name}: input {ti:6.1f} us, output {to:6.1f} us, grad-output {tg:6.1f} us", flush=True)
