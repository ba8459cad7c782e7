"""Roofline table of the HBM-bound hand-written kernels (K5 variants, K6 transforms, K6f, K7, K8) at
the shapes of the BASELINE workload (37 images / 64 clouds): time per call, ALGORITHMIC bytes per
call (stated per kernel in DESIGN.md section 3) and the resulting GB/s against the 8 TB/s peak.

    python tools/bench_streams.py > profiles/rNN/kernel_table_streams.txt      # on the GPU box
"""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpsg_amd import winograd as wg  # noqa: E402
from fpsg_amd.conv_first import conv3x3_first  # noqa: E402
from fpsg_amd.fused_bn import bn_act, conv_bn_act_max, conv_bn_act_pool  # noqa: E402
from fpsg_amd.optim import FlatAdam  # noqa: E402

HBM_PEAK = 8.0e12
dev = torch.device("cuda:0")


def gpu_time(fn, reps=20, warm=3):
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


def row(name, shape, t, nbytes, note=""):
    print(f"{name:34s} {shape:22s} {t * 1e6:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / t / 1e9:8.0f} GB/s  "
          f"{100 * nbytes / t / HBM_PEAK:5.1f} %  {note}")


def main():
    print("# kernel                             shape                       time     alg. bytes      rate   of 8 TB/s")
    mb = lambda *dims: 4 * int(torch.tensor(dims).prod())
    # ---- K5: BatchNorm + ReLU, sliced path (forward 2R+1W, backward 4R+1W of the tensor)
    for shape in ((37, 64, 224, 224), (37, 128, 112, 112), (37, 256, 56, 56), (37, 512, 28, 28)):
        bn = nn.BatchNorm2d(shape[1]).to(dev).train()
        x = torch.randn(*shape, device=dev, requires_grad=True)
        g = torch.randn(*shape, device=dev)
        t_f = gpu_time(lambda: bn_act(bn, x, "relu"))
        y = bn_act(bn, x, "relu")
        t_b = gpu_time(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        row("K5 bn+relu forward", str(shape), t_f, 3 * mb(*shape))
        row("K5 bn+relu backward", str(shape), t_b, 5 * mb(*shape))
        del x, g, y
    # ---- K5 pooled (forward 2R + W/4, backward 2R + 2(R/4) + 1W)
    for shape in ((37, 64, 224, 224), (37, 256, 56, 56)):
        C = shape[1]
        conv = nn.Conv2d(C, C, 1).to(dev)
        with torch.no_grad():
            conv.weight.copy_(torch.eye(C).view(C, C, 1, 1))
        bn, pool = nn.BatchNorm2d(C).to(dev).train(), nn.MaxPool2d(2, 2)
        x = torch.randn(*shape, device=dev)
        from fpsg_amd.fused_bn import _BNActPool
        xr = x.clone().requires_grad_()
        fn = lambda: _BNActPool.apply(xr, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, 1, 0.0,
                                      conv.bias, 0.1)
        t_f = gpu_time(fn)
        yp = fn()
        gp = torch.randn_like(yp)
        t_b = gpu_time(lambda: torch.autograd.grad(yp, xr, gp, retain_graph=True))
        row("K5 bn+relu+pool forward", str(shape), t_f, int(2.25 * mb(*shape)))
        row("K5 bn+relu+pool backward", str(shape), t_b, int(3.5 * mb(*shape)))
        del x, xr, yp, gp
    # ---- K5 max over points (forward 1R, backward 1R+1W)
    shape = (64, 1024, 2048)
    bn = nn.BatchNorm1d(1024).to(dev).train()
    from fpsg_amd.fused_bn import _BNActMax
    xr = torch.randn(*shape, device=dev, requires_grad=True)
    pb = torch.zeros(1024, device=dev)
    fn = lambda: _BNActMax.apply(xr, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, 1, 0.0, pb, 0.1)
    t_f = gpu_time(fn)
    out = fn()
    go = torch.randn_like(out)
    t_b = gpu_time(lambda: torch.autograd.grad(out, xr, go, retain_graph=True))
    row("K5 bn+relu+max forward", str(shape), t_f, mb(*shape))
    row("K5 bn+relu+max backward", str(shape), t_b, 2 * mb(*shape))
    del xr, out
    # ---- K6 transforms (m = 4): input 1R + 2.25W, output 2.25R + 1W, grad-output 1R + 2.25W
    for (C, H) in ((128, 112), (256, 56), (512, 28)):
        x = torch.randn(37, C, H, H, device=dev)
        V = wg._input(4, x)
        t_in = gpu_time(lambda: wg._input(4, x))
        t_out = gpu_time(lambda: wg._output(4, V, 37, H, H))
        t_go = gpu_time(lambda: wg._grad_output(4, x))
        row("K6 input transform (m=4)", f"(37, {C}, {H}, {H})", t_in, int(3.25 * x.numel() * 4))
        row("K6 output transform (m=4)", f"(37, {C}, {H}, {H})", t_out, int(3.25 * x.numel() * 4))
        row("K6 grad-output transform (m=4)", f"(37, {C}, {H}, {H})", t_go, int(3.25 * x.numel() * 4))
        del x, V
    # ---- K6f (64 input channels, one kernel): algorithmic bytes = input + output tensors
    for (K, H) in ((64, 224), (128, 112)):
        x = torch.randn(37, 64, H, H, device=dev)
        w = torch.randn(K, 64, 3, 3, device=dev) * 0.05
        U = wg._filter(4, w, False)
        t = gpu_time(lambda: wg._fused(x, U))
        flops = 2 * 36 * K * 64 * 37 * (H // 4) ** 2
        row("K6f fused conv 64->%d" % K, f"(37, 64, {H}, {H})", t, (x.numel() + 37 * K * H * H) * 4,
            f"{flops / t / 1e12:.0f} TFLOP/s in the Winograd domain (x4 direct-equivalent)")
        del x
    # ---- K7 Adam over the full model's 77.4 M parameters (4R + 3W)
    n = 77445125
    p = nn.Parameter(torch.randn(n, device=dev))
    opt = FlatAdam([p], lr=1e-3)
    p.grad = torch.randn(n, device=dev)
    opt.bind_gradients(p.grad)
    t = gpu_time(lambda: opt.step())
    row("K7 Adam step", f"({n},)", t, 28 * n)
    del p, opt
    # ---- K8 first-layer weight gradient (dy read once)
    x = torch.randn(37, 3, 224, 224, device=dev)
    w = torch.randn(64, 3, 3, 3, device=dev, requires_grad=True)
    g = torch.randn(37, 64, 224, 224, device=dev)
    y = conv3x3_first(x, w)
    t = gpu_time(lambda: torch.autograd.grad(y, w, g, retain_graph=True))
    row("K8 first-layer dw", "(37, 64, 224, 224)", t, g.numel() * 4 + x.numel() * 4)


if __name__ == "__main__":
    main()
