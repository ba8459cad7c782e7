"""K6f (wino4_fused_c64_kernel) alone at the episode's two shapes: time, Winograd-domain TFLOP/s against the
157.3 TFLOP/s fp32 MFMA peak, and the largest difference from the three-kernel K6 form.
    python tools/bench_k6f.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import winograd as wg  # noqa: E402

dev = torch.device("cuda:0")


def gpu_time(fn, reps=20, warm=3):
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


torch.manual_seed(0)
for (K, H) in ((64, 224), (128, 112)):
    x = torch.randn(37, 64, H, H, device=dev)
    w = torch.randn(K, 64, 3, 3, device=dev) * 0.05
    U = wg._filter(4, w, False)
    y = wg._fused(x, U)
    ref = wg._output(4, torch.bmm(U, wg._input(4, x)), 37, H, H)
    err = (y - ref).abs().max().item()
    t = gpu_time(lambda: wg._fused(x, U))
    flops = 2 * 36 * K * 64 * 37 * (H // 4) ** 2
    print(f"K6f 64->{K} (37,64,{H},{H}) {t * 1e6:8.1f} us  {flops / t / 1e12:6.1f} TFLOP/s (Winograd domain)  "
          f"{100 * flops / t / 157.3e12:5.1f} % of fp32 MFMA peak   max |K6f - K6| = {err:.2e}")

# K6w: the weight gradient of the same layers in one pass, against the three-kernel form it replaces
for (K, H) in ((64, 224), (128, 112)):
    x = torch.randn(37, 64, H, H, device=dev)
    gy = torch.randn(37, K, H, H, device=dev)
    three = lambda: torch.bmm(wg._grad_output(4, gy), wg._input(4, x).transpose(1, 2))
    ref = three()
    dU = wg._fused_dw(x, None, None, gy)
    err = (dU - ref).abs().max().item() / ref.abs().max().item()
    t3 = gpu_time(three, reps=10)
    t = gpu_time(lambda: wg._fused_dw(x, None, None, gy), reps=10)
    flops = 2 * 36 * K * 64 * 37 * (H // 4) ** 2
    print(f"K6w dU 64->{K} (37,64,{H},{H}) {t * 1e6:8.1f} us  {flops / t / 1e12:6.1f} TFLOP/s (Winograd domain)  "
          f"{100 * flops / t / 157.3e12:5.1f} % of fp32 MFMA peak   three-kernel form {t3 * 1e6:8.1f} us   "
          f"max |K6w - form| / scale = {err:.2e}")
    del x, gy, ref, dU
