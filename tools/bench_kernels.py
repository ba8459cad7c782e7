"""Kernel table of SURVEY.md 8(d): every hand-written HIP kernel at the BASELINE shapes, with
its roofline figures, next to the same math on the host cores (the C oracle on one core and a
pure-PyTorch restatement on all usable cores; the reference itself has no CPU path).

    python tools/bench_kernels.py            # on the GPU box
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (baseline leg only)
from bench import usable_cores  # noqa: E402
from fpsg_amd import _hip  # noqa: E402
from fpsg_amd.dgcnn import knn_int32  # noqa: E402
from fpsg_amd.metrics import emd_approx  # noqa: E402

F32_PEAK, HBM_PEAK = 157.3e12, 8.0e12


def gpu_time(fn, reps, warm=5):
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


def cpu_time(fn, min_s=1.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min_s:
        fn()
        n += 1
    return (time.perf_counter() - t0) / n


def torch_cpu_chamfer(a, b):
    d = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)
    return d.min(2)[0].mean(1) + d.min(1)[0].mean(1)


def torch_cpu_knn(x, k):      # formula of reference src/dgcnn/model.py:13-20
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]


def main():
    dev = torch.device("cuda:0")
    cores = usable_cores()
    torch.set_num_threads(cores)
    lib = _hip.load()
    N = 2048
    print(f"# host: {cores} usable cores; GPU: {torch.cuda.get_device_name(0)}")
    print("# K1 Chamfer (N=M=2048): algorithmic 81,920 B and 2*N*M*8 = 67.1 MFLOP per cloud pair (fwd)")
    print(f"{'kernel':10s} {'B':>5s} {'us':>10s} {'TFLOP/s':>9s} {'%f32peak':>9s} {'GB/s':>8s} {'%HBM':>6s}")
    for B in (1, 5, 32, 37, 256, 2048):
        p1 = torch.rand(B, N, 3, device=dev) * 2 - 1
        p2 = torch.tanh(torch.randn(B, N, 3, device=dev))
        d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
        i1 = torch.empty(B, N, device=dev, dtype=torch.int32); i2 = torch.empty_like(i1)
        g1 = torch.randn(B, N, device=dev); g2 = torch.randn(B, N, device=dev)
        gx1 = torch.empty_like(p1); gx2 = torch.empty_like(p2)
        s = torch.cuda.current_stream().cuda_stream
        tf = gpu_time(lambda: lib.fpsg_chamfer_fwd(p1.data_ptr(), p2.data_ptr(), B, N, N, d1.data_ptr(), i1.data_ptr(),
                                                   d2.data_ptr(), i2.data_ptr(), s), 100 if B < 512 else 20)
        tb = gpu_time(lambda: lib.fpsg_chamfer_bwd(p1.data_ptr(), p2.data_ptr(), i1.data_ptr(), i2.data_ptr(),
                                                   g1.data_ptr(), g2.data_ptr(), B, N, N, gx1.data_ptr(), gx2.data_ptr(), s),
                      100 if B < 512 else 20)
        fl = B * 2.0 * N * N * 8
        print(f"{'K1 fwd':10s} {B:5d} {tf*1e6:10.2f} {fl/tf/1e12:9.2f} {fl/tf/F32_PEAK*100:9.1f} {B*81920/tf/1e9:8.1f} {B*81920/tf/HBM_PEAK*100:6.2f}")
        print(f"{'K1 bwd':10s} {B:5d} {tb*1e6:10.2f} {'':9s} {'':9s} {B*131072/tb/1e9:8.1f} {B*131072/tb/HBM_PEAK*100:6.2f}")
    # CPU legs on a bounded sample
    a = (torch.rand(4, N, 3) * 2 - 1); b = torch.tanh(torch.randn(4, N, 3))
    t_or = cpu_time(lambda: oracle.chamfer_fwd(a.numpy(), b.numpy()))
    t_th = cpu_time(lambda: torch_cpu_chamfer(a, b))
    print(f"# CPU, 4 cloud pairs: C oracle (1 core) {t_or/4*1e3:.1f} ms/pair = {4/t_or:.1f} pairs/s; "
          f"PyTorch broadcast+min ({cores} cores) {t_th/4*1e3:.1f} ms/pair = {4/t_th:.1f} pairs/s")

    print("\n# K2 approximate-assignment EMD (N=M=2048): 30 sweeps of N*M pairs per cloud pair (forward only)")
    for B in (1, 5):
        p1 = torch.rand(B, N, 3, device=dev) * 2 - 1
        p2 = torch.tanh(torch.randn(B, N, 3, device=dev))
        t = gpu_time(lambda: emd_approx(p1, p2), 10, 2)
        print(f"K2 emd     B={B}: {t*1e3:8.3f} ms   {B*30*N*N/t/1e12:6.3f} Tpair-sweeps/s")
    from fpsg_amd.metrics import sinkhorn_divergence
    for B in (1, 5):
        p1 = torch.rand(B, N, 3, device=dev) * 2 - 1
        p2 = torch.tanh(torch.randn(B, N, 3, device=dev))
        t = gpu_time(lambda: sinkhorn_divergence(p1, p2, diameter=3.5), 5, 2)
        print(f"K2b sinkhorn divergence B={B}: {t*1e3:8.3f} ms  (~40 soft-mins of N*M pairs)")
    a1 = (torch.rand(1, N, 3) * 2 - 1).numpy(); b1 = torch.tanh(torch.randn(1, N, 3)).numpy()
    t0 = time.perf_counter(); oracle.emd_approx(a1, b1); t_or = time.perf_counter() - t0
    print(f"# CPU: C oracle (1 core) {t_or*1e3:.0f} ms per cloud pair")

    print("\n# K3 kNN graph (B=64, N=2048, k=20): 2*C*N*N flop per cloud in the x^T x term")
    for C in (3, 64, 128):
        x = torch.randn(64, C, N, device=dev)
        t = gpu_time(lambda: knn_int32(x, 20), 10, 2)
        print(f"K3 knn     C={C:3d}: {t*1e3:8.3f} ms   {64*2.0*C*N*N/t/1e12:6.2f} TFLOP/s   {64*N*N/t/1e12:6.3f} Tpair/s")
        xc = torch.randn(8, C, N)
        t_th = cpu_time(lambda: torch_cpu_knn(xc, 20))
        print(f"#   CPU PyTorch knn formula ({cores} cores), B=8: {t_th*1e3:.1f} ms  -> {t_th/8*64*1e3:.0f} ms per 64 clouds")


if __name__ == "__main__":
    main()
