"""K5 (BatchNorm + ReLU) forward / backward at PointNet's and one trunk shape: small tensors are launch-latency bound."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from fpsg_amd.fused_bn import bn_act
dev = "cuda"
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for shape in ((64, 64, 2048), (64, 128, 2048), (37, 64, 2048), (37, 128, 112, 112), (37, 64, 224, 224), (64, 1024, 2048)):
    C = shape[1]
    bn = (nn.BatchNorm1d(C) if len(shape) == 3 else nn.BatchNorm2d(C)).to(dev).train()
    x = torch.randn(*shape, device=dev, requires_grad=True)
    g = torch.randn(*shape, device=dev)
    y = bn_act(bn, x, "relu")
    tf = timed(lambda: bn_act(bn, x, "relu"))
    tb = timed(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    mb = x.numel() * 4 / 1e6
    print(shape, "fwd", round(tf, 1), "us", round(3 * mb / tf, 2), "TB/s*1e-3  bwd", round(tb, 1), "us", round(5 * mb / tb / 1e3, 2), "TB/s")
