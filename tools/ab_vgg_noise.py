"""A/B of the image trunk's gradient noise: distance of every variant's parameter gradients from a
float64 CPU run of the same trunk (6 images, training-mode BatchNorm), per variant.
    python tools/ab_vgg_noise.py   (on the GPU box)"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd.image_net import ImageEncoderWarpper

torch.manual_seed(1)
enc = ImageEncoderWarpper("vgg_16").train()
x = torch.rand(6, 3, 224, 224) * 2 - 1
w = torch.randn(6, 512)
e64 = copy.deepcopy(enc).double()
t0 = time.time()
(e64(x.double()) * w.double()).sum().backward()
g64 = torch.cat([p.grad.reshape(-1) for p in e64.parameters()])
print("float64 CPU", time.time() - t0, "s")
e32 = copy.deepcopy(enc)
(e32(x) * w).sum().backward()
g32 = torch.cat([p.grad.reshape(-1) for p in e32.parameters()]).double()
print("cpu fp32 (reference arithmetic)", float((g32 - g64).norm() / g64.norm()))
gpu = torch.device("cuda:0")
for label, env in (("default (K5+K6+K6f+K8)", {}), ("FPSG_WINOGRAD=0", {"FPSG_WINOGRAD": "0"}),
                   ("FPSG_FUSED_BN=0", {"FPSG_FUSED_BN": "0"}), ("both off (library only)", {"FPSG_WINOGRAD": "0", "FPSG_FUSED_BN": "0"}),
                   ("FPSG_WINOGRAD_M=2", {"FPSG_WINOGRAD_M": "2"}), ("FPSG_WINOGRAD_FUSED=0", {"FPSG_WINOGRAD_FUSED": "0"})):
    for k in ("FPSG_WINOGRAD", "FPSG_FUSED_BN", "FPSG_WINOGRAD_M", "FPSG_WINOGRAD_FUSED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eg = copy.deepcopy(enc).to(gpu)
    out = eg(x.to(gpu))
    (out * w.to(gpu)).sum().backward()
    gg = torch.cat([p.grad.reshape(-1) for p in eg.parameters()]).double().cpu()
    f64out = e64(x.double()).detach()
    print(f"{label:28s} grad L2 dev {float((gg - g64).norm() / g64.norm()):.4e}   fwd max dev {float((out.detach().cpu().double() - f64out).abs().max() / f64out.abs().max()):.3e}")
