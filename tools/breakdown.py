"""Where an episode step spends its time: per-module forward / forward+backward timings
(HIP events) at the config-3 shapes (32-shot, 5-query).  Usage: python tools/breakdown.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd.engine import build_model, build_optimizer, default_options  # noqa: E402
from fpsg_amd.metrics import chamfer_distance  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fb(module_fn, *inputs):
    def run():
        outs = module_fn(*inputs)
        outs.sum().backward()
    return run


def main():
    dev = torch.device("cuda:0")
    enc = sys.argv[1] if len(sys.argv) > 1 else "pointnet"
    opt = default_options(device="cuda", intra_recon=True, pc_encoder=enc)
    torch.manual_seed(0)
    m = build_model(opt).to(dev).train()
    S, Q = 32, 5
    img = torch.rand(S + Q, 3, 224, 224, device=dev) * 2 - 1
    pc = torch.rand(2 * S, 3, 2048, device=dev) * 2 - 1
    lat5 = torch.randn(Q, 1536, device=dev, requires_grad=True)
    lat32 = torch.randn(S, 1536, device=dev, requires_grad=True)
    gt5 = torch.rand(Q, 2048, 3, device=dev)
    gt32 = torch.rand(S, 2048, 3, device=dev)
    with torch.no_grad():
        t_vgg_f = timeit(lambda: m.img_encoder(img))
        t_pc_f = timeit(lambda: m.pc_encoder(pc))
        t_d5_f = timeit(lambda: m.pc_decoder(lat5))
        t_d32_f = timeit(lambda: m.pc_decoder(lat32))
    t_vgg = timeit(fb(m.img_encoder, img))
    t_pc = timeit(fb(m.pc_encoder, pc))
    t_d5 = timeit(fb(m.pc_decoder, lat5))
    t_d32 = timeit(fb(m.pc_decoder, lat32))
    syn = torch.tanh(torch.randn(S, 2048, 3, device=dev)).requires_grad_()
    t_cd = timeit(fb(lambda a, b: chamfer_distance(a, b), syn, gt32))
    optimizer, _ = build_optimizer(m, opt)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    t_opt = timeit(lambda: optimizer.step())
    print(f"encoder={enc}  (ms)           fwd     fwd+bwd")
    print(f"VGG16-BN   [37,3,224,224]   {t_vgg_f:7.2f}  {t_vgg:8.2f}")
    print(f"pc encoder [64,3,2048]      {t_pc_f:7.2f}  {t_pc:8.2f}")
    print(f"decoder    B=5              {t_d5_f:7.2f}  {t_d5:8.2f}")
    print(f"decoder    B=32             {t_d32_f:7.2f}  {t_d32:8.2f}")
    print(f"chamfer    B=32 (+mean)              {t_cd:8.2f}")
    print(f"Adam (fused) 77.4 M                  {t_opt:8.2f}")
    print(f"sum fwd+bwd                          {t_vgg + t_pc + t_d5 + t_d32 + t_cd + t_opt:8.2f}")


if __name__ == "__main__":
    main()
