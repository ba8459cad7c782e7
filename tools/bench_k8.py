"""K8 (first-layer weight gradient) alone at the episode's shape: the plain form and the form with the BatchNorm backward
folded in, GB/s of the tensors they read.  Usage (GPU box): [FPSG_K8_PX=224] python tools/bench_k8.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
lib = _hip.load()
N, H, W = 37, 224, 224
x = torch.randn(N, 3, H, W, device=dev)
y = torch.randn(N, 64, H, W, device=dev)
ga = torch.randn(N, 64, H, W, device=dev)
chan = torch.randn(4, 64, device=dev)
coef = torch.randn(3, 64, device=dev)
pb = torch.randn(64, device=dev)
dw = torch.empty(64, 3, 3, 3, device=dev)
ws = torch.empty(lib.fpsg_conv_first_dw_workspace_floats(N, H, W), device=dev)
st = torch.cuda.current_stream().cuda_stream
warm = torch.randn(4096, 4096, device=dev)
for _ in range(50):
    warm = torch.tanh(warm)


def t(fn, reps=30):
    for _ in range(3):
        assert fn() == 0, lib.fpsg_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


plain = t(lambda: lib.fpsg_conv_first_dw(x.data_ptr(), ga.data_ptr(), N, 3, 64, H, W, dw.data_ptr(), ws.data_ptr(), st))
ref = dw.clone()
fold = t(lambda: lib.fpsg_conv_first_dw_fold(x.data_ptr(), y.data_ptr(), ga.data_ptr(), chan.data_ptr(), coef.data_ptr(),
                                             pb.data_ptr(), N, 3, 64, H, W, dw.data_ptr(), ws.data_ptr(), st))
b1 = (ga.numel() + x.numel()) * 4
b2 = (ga.numel() + y.numel() + x.numel()) * 4
print(f"FPSG_K8_PX={os.environ.get('FPSG_K8_PX', '-')}: plain {plain * 1e6:7.1f} us {b1 / plain / 1e9:7.0f} GB/s ({b1 / plain / 8e12:.3f})   "
      f"folded {fold * 1e6:7.1f} us {b2 / fold / 1e9:7.0f} GB/s ({b2 / fold / 8e12:.3f})   checksum {float(ref.double().sum()):.6e}")
