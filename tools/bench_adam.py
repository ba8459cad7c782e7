"""K7 alone: FlatAdam.step() over the model's 77.4 M parameters with the gradient bound as one flat buffer (adam_kernel)."""
import sys, torch
sys.path.insert(0, ".")
import torch.nn as nn
from fpsg_amd.optim import FlatAdam
n = 77445125
p = nn.Parameter(torch.randn(n, device="cuda"))
opt = FlatAdam([p], lr=1e-3)
p.grad = torch.randn(n, device="cuda")
opt.bind_gradients(p.grad)
for _ in range(3): opt.step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): opt.step()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
print("adam_kernel", t * 1e6, "us", 28.0 * n / t / 1e9, "GB/s")
