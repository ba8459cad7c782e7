"""K6 vs the library convolution at the VGG16 layer shapes of an episode (37 images): forward,
data gradient, weight gradient.  python tools/bench_winograd.py [N]"""
import sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from fpsg_amd import winograd as wg

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 37


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"# N={N}; ms per call; lib = F.conv2d / torch.nn.grad (MIOpen), wino = K6 transforms + torch.bmm")
print(f"# {'layer':22s} {'lib fwd':>8s} {'wino fwd':>9s} | {'lib dx':>8s} {'wino dx':>8s} | {'lib dw':>8s} {'wino dw':>8s} | parts: in-T, gemm, out-T")
for (C, K, H) in [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14)]:
    x = torch.randn(N, C, H, H, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    gy = torch.randn(N, K, H, H, device=dev)
    l_f = timeit(lambda: F.conv2d(x, w, None, 1, 1))
    l_dx = timeit(lambda: torch.nn.grad.conv2d_input(x.shape, w, gy, padding=1))
    l_dw = timeit(lambda: torch.nn.grad.conv2d_weight(x, w.shape, gy, padding=1))
    for m in ((2, 4) if H % 4 == 0 else (2,)):
        w_f = timeit(lambda: wg._output(m, torch.bmm(wg._filter(m, w, False), wg._input(m, x)), N, H, H))
        w_dx = timeit(lambda: wg._output(m, torch.bmm(wg._filter(m, w, True), wg._input(m, gy)), N, H, H))
        V = wg._input(m, x)
        P = V.shape[2]
        w_dw = timeit(lambda: wg._filter_grad(m, torch.bmm(wg._grad_output(m, gy), V.transpose(1, 2)), w))
        U = wg._filter(m, w, False)
        t_in = timeit(lambda: wg._input(m, x)); t_g = timeit(lambda: torch.bmm(U, V)); M = torch.bmm(U, V)
        t_out = timeit(lambda: wg._output(m, M, N, H, H))
        if m == 4 and C == 64:
            t_f = timeit(lambda: wg._fused(x, U))
            err = float((wg._fused(x, U) - wg._output(m, M, N, H, H)).abs().max() / M.abs().max())
            print(f"    fused one-kernel forward (K6f): {t_f:.3f} ms   max |diff| vs three-kernel form / scale = {err:.2e}", flush=True)
        print(f"{C:4d}->{K:4d} @{H:3d}x{H:<3d} m={m}   {l_f:8.3f} {w_f:9.3f} | {l_dx:8.3f} {w_dx:8.3f} | {l_dw:8.3f} {w_dw:8.3f} | "
              f"{t_in:.3f} ({(x.numel()+V.numel())*4/t_in/1e9:.1f} TB/s), {t_g:.3f} ({2*(m+2)**2*K*C*P/t_g/1e9:.0f} TF/s), "
              f"{t_out:.3f} ({(M.numel()+gy.numel())*4/t_out/1e9:.1f} TB/s)", flush=True)
        del V, M
