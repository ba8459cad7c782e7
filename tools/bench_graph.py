"""K3 / K4b micro-benchmark: kNN graph and fused EdgeConv at DGCNN's layer shapes
(B = 64 clouds = 2*n_shot, N = 2048, k = 20).  Usage (GPU box): python tools/bench_graph.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd.dgcnn import DGCNNfeat, edgeconv_fused, knn_int32  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = torch.device("cuda:0")
    B, N, k = 64, 2048, 20
    net = DGCNNfeat().to(dev).train()
    for C, block in ((3, net.conv1), (64, net.conv2), (64, net.conv3), (128, net.conv4)):
        x = torch.randn(B, C, N, device=dev)
        t = timeit(lambda: knn_int32(x, k))
        flop = 2.0 * C * N * N * B
        print(f"knn   B={B} C={C:3d} N={N}: {t*1e3:8.3f} ms  {flop/t/1e12:6.2f} TFLOP/s (xTx only)  "
              f"{B*N*N/t/1e12:6.3f} Tpair/s", flush=True)
        idx = knn_int32(x, k)
        xpm = x.transpose(1, 2).contiguous().requires_grad_()
        Co = block[0].out_channels

        def fwd():
            return edgeconv_fused(xpm, idx, block)

        tf = timeit(fwd)
        out = fwd()
        g = torch.randn_like(out)

        def fwdbwd():
            o = edgeconv_fused(xpm, idx, block)
            o.backward(g)

        tb = timeit(fwdbwd)
        gather = B * N * k * Co * 4
        print(f"edgeconv C={C:3d}->Co={Co:3d}: fwd {tf*1e3:7.3f} ms  fwd+bwd {tb*1e3:7.3f} ms   "
              f"gather {gather/1e9:5.2f} GB -> {gather/tf/1e12:5.2f} TB/s (fwd, incl. GEMM + stats)", flush=True)
    x = torch.randn(B, 3, N, device=dev)
    t = timeit(lambda: net(x), reps=5)
    print(f"DGCNNfeat fwd (train) B={B}: {t*1e3:.2f} ms")
    net_u = DGCNNfeat(fused=False).to(dev).train()
    xs = x[:16].contiguous()
    tu = timeit(lambda: net_u(xs), reps=3)
    tfu = timeit(lambda: net(xs), reps=3)
    print(f"DGCNNfeat fwd B=16: fused {tfu*1e3:.2f} ms vs reference chain (materialised [B,2C,N,k]) {tu*1e3:.2f} ms")


if __name__ == "__main__":
    main()
