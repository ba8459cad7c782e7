"""K2 / K2b micro-benchmark at the evaluation shapes (Q = 1 and 5 query clouds of 2048 points).
Usage (GPU box): python tools/bench_emd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd.metrics import emd_approx, sinkhorn_divergence, sinkhorn_epsilons  # noqa: E402

dev = torch.device("cuda:0")
for B in (1, 5):
    x = torch.rand(B, 2048, 3, device=dev) * 2 - 1
    y = torch.tanh(torch.randn(B, 2048, 3, device=dev))
    n_eps = len(sinkhorn_epsilons(3.5))
    for name, fn, sweeps in (("K2b sinkhorn_divergence", lambda: sinkhorn_divergence(x, y, diameter=3.5), 4 * (n_eps + 2)),
                             ("K2 emd_approx (forward)", lambda: emd_approx(x, y), 30)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        print(f"{name:26s} B={B}: {t*1e3:8.3f} ms   {sweeps} sweeps of N*M pairs -> {B*sweeps*2048*2048/t/1e12:6.3f} Tpair/s")
