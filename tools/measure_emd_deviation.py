"""Measured deviation of the HIP EMD kernels (K2 approximate assignment, K2b Sinkhorn divergence) from their CPU
restatements at the shapes of tests/test_emd_gpu.py -- the numbers the tests' tolerances are set from (3x, rounded).
Usage (GPU box): python tools/measure_emd_deviation.py > profiles/r03/emd_deviation.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import unit_ball_clouds  # noqa: E402
from fpsg_amd.metrics import emd_approx, sinkhorn_divergence  # noqa: E402

gpu = torch.device("cuda:0")


def clouds(seed, B, N, M):
    rng = np.random.default_rng(seed)
    return unit_ball_clouds(rng, B, N), np.tanh(rng.standard_normal((B, M, 3)) * 0.5).astype(np.float32)


print("# K2 emd_approx: HIP vs oracle.emd_approx -- max relative cost deviation; gradient deviation / max |gradient|")
for B, N, M in [(2, 256, 256), (1, 1024, 1024), (3, 100, 300), (2, 512, 128), (1, 2048, 2048)]:
    for seed in (N + M, N + M + 1, N + M + 2):
        p1, p2 = clouds(seed, B, N, M)
        t1 = torch.from_numpy(p1).to(gpu).requires_grad_()
        t2 = torch.from_numpy(p2).to(gpu).requires_grad_()
        cost = emd_approx(t1, t2)
        cost.sum().backward()
        oc, og1, og2 = oracle.emd_approx(p1, p2, want_grad=True)
        dc = np.abs(cost.detach().cpu().numpy() - oc) / np.abs(oc)
        d1 = np.abs(t1.grad.cpu().numpy() - og1).max() / np.abs(og1).max()
        d2 = np.abs(t2.grad.cpu().numpy() - og2).max() / np.abs(og2).max()
        print(f"B={B} N={N} M={M} seed={seed}: cost {dc.max():.3e}  grad1 {d1:.3e}  grad2 {d2:.3e}", flush=True)

print("# K2b sinkhorn_divergence: HIP vs oracle.sinkhorn_divergence (fp32 C restatement) -- max relative deviation")
for B, N, M, seed in [(3, 512, 512, 8), (3, 512, 512, 9), (2, 700, 600, 31), (1, 2048, 2048, 5), (2, 300, 1000, 6)]:
    rng = np.random.default_rng(seed)
    x = unit_ball_clouds(rng, B, N)
    y = (unit_ball_clouds(rng, B, M) * 0.7 + 0.2).astype(np.float32)
    got = sinkhorn_divergence(torch.from_numpy(x).to(gpu), torch.from_numpy(y).to(gpu)).cpu().numpy()
    exp = oracle.sinkhorn_divergence(x, y)
    print(f"B={B} N={N} M={M} seed={seed}: {np.abs(got - exp).max() / np.abs(exp).max():.3e}  (values {got[:2]})", flush=True)
