"""The first convolution of the VGG16-BN trunk (3 -> 64 channels, 3x3, padding 1;
``vgg16_bn.features[0]``, reference ``src/models/image_net.py:14``).

Forward (and the data gradient, which the train step never asks for: images carry no gradient) are
the library's; the WEIGHT gradient is K8 (``fpsg_conv_first_dw``): ``dy`` -- 475 MB at 37 images --
is read once, where the library spends two layout transposes and an NHWC implicit GEMM on a
1,728-element result."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import _hip


def eligible(x: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    def is_(v, want):
        return v == want or v == (want, want)
    return (os.environ.get("FPSG_CONV_FIRST", "1") != "0" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and conv.in_channels == 3 and conv.out_channels == 64 and is_(conv.kernel_size, 3) and is_(conv.stride, 1)
            and is_(conv.padding, 1) and is_(conv.dilation, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and x.shape[3] % 4 == 0)


class _ConvFirst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.conv2d(x, w, None, 1, 1)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.nn.grad.conv2d_input(x.shape, w, gy, padding=1)
        if ctx.needs_input_grad[1]:
            gy = gy.contiguous()
            xc = x.contiguous()
            N, _, H, W = xc.shape
            lib = _hip.load()
            gw = torch.empty_like(w, memory_format=torch.contiguous_format)
            ws = torch.empty((lib.fpsg_conv_first_dw_workspace_floats(N, H, W),), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                rc = lib.fpsg_conv_first_dw(_hip.ptr(xc), _hip.ptr(gy), N, 3, 64, H, W, _hip.ptr(gw), _hip.ptr(ws),
                                            _hip.stream_of(gy))
            _hip.check(rc, "fpsg_conv_first_dw")
        return gx, gw


def conv3x3_first(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``F.conv2d(x, weight, None, 1, 1)`` for ``x [N,3,H,W]``, ``weight [64,3,3,3]``."""
    return _ConvFirst.apply(x, weight)
