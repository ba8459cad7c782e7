"""K2 forward-only variants (fpsg_emd_approx_variant) at the evaluation shapes, back to back.
Usage (GPU box): python tools/bench_emd_variants.py > profiles/r04/k2_variants.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
lib = _hip.load()
warm = torch.randn(4096, 4096, device=dev)
for _ in range(100):
    warm = torch.tanh(warm)
print("# variant bits: 1 = assignment + next level's row normalisers in one launch, 2 = four owner points per wave")
for B in (1, 5, 37):
    x = torch.rand(B, 2048, 3, device=dev) * 2 - 1
    y = torch.tanh(torch.randn(B, 2048, 3, device=dev))
    ws = torch.empty((lib.fpsg_emd_workspace_floats(B, 2048, 2048),), dtype=torch.float32, device=dev)
    cost = torch.empty((B,), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ref = None
    for variant in (0, 1, 2, 3):
        fn = lambda: lib.fpsg_emd_approx_variant(x.data_ptr(), y.data_ptr(), B, 2048, 2048, cost.data_ptr(), None, None,
                                                 ws.data_ptr(), variant, st)
        for _ in range(5):
            assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 30 * 1e-3
        c = cost.cpu()
        ref = c if ref is None else ref
        flop = B * 2048.0 * 2048 * 10 * (2 * 12 + 14)
        print(f"B={B:3d} variant {variant}: {t * 1e6:8.1f} us   {flop / t / 1e12:6.2f} TFLOP/s ({flop / t / 157.3e12:.3f} of the fp32 peak, "
              f"30-sweep flop count)   max rel. difference from variant 0: {float(((c - ref).abs() / ref.abs()).max()):.1e}")
