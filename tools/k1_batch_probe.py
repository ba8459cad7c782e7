import torch, sys, time
sys.path.insert(0, '/root/repo')
from fpsg_amd import metrics as m
dev = torch.device('cuda:0')
def ev(fn, reps, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(7)
for B in (1, 5, 32, 33, 37, 32, 64, 256):
    p1 = (torch.rand(B, 2048, 3, generator=g) * 2 - 1).to(dev)
    p2 = torch.tanh(torch.randn(B, 2048, 3, generator=g)).to(dev)
    t = ev(lambda: m._sided_forward(p1, p2), 200)
    t0 = time.perf_counter()
    for _ in range(200): m._sided_forward(p1, p2)
    torch.cuda.synchronize()
    print(B, round(t, 1), 'us (events)', round((time.perf_counter() - t0) / 200 * 1e6, 1), 'us (wall)', flush=True)
