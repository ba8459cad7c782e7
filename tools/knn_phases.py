import sys, torch
sys.path.insert(0, ".")
from fpsg_amd.dgcnn import knn_int32
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for C in (3, 64, 128):
    x = torch.randn(64, C, 2048, device=dev)
    print(C, {k: round(t(lambda: knn_int32(x, k)), 3) for k in (1, 5, 10, 20, 40)})
