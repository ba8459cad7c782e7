"""3x3 convolution through Winograd F(2x2, 3x3) (K6) for the deep layers of the image trunk.

``conv3x3(x, weight)`` is ``F.conv2d(x, weight, None, stride=1, padding=1)`` with gradients to
both arguments.  The 16 transform-domain GEMMs are fp32 batched matrix products on the MFMA
pipes (``torch.bmm`` = hipBLASLt); the input / output / filter transforms are the HIP kernels
of ``csrc/winograd.hip`` through the C ABI.  2.25x fewer multiplications than the direct form;
measured on MI355X against MIOpen's own fp32 Winograd kernel (the solver it picks for these
layers) this is faster from 256 channels up (weight gradient: from 128), where the 4x larger transform-domain tensors are
small next to the GEMM work (``profiles/``: wino_*).  Reference layers: the Conv2d(3x3, pad 1) of
torchvision's ``vgg16_bn.features`` built at ``src/models/image_net.py:14``.

The forward keeps the transformed input ``V`` (not ``x``) for the weight gradient
``dU = dM V^T``; with 288 GB of HBM the 4x larger tensor (<= 475 MB per layer here) is cheap.
"""
from __future__ import annotations

import os

import torch

from . import _hip

# Below these widths the transforms' HBM traffic (4x the image tensors) outweighs the saved
# multiplications (profiles/: wino_bench): min(C,K) >= 128 and max(C,K) >= 256 selects conv3_1 ...
# conv5_3 of VGG16, nine of its thirteen layers.
MIN_CHANNELS = 128
MIN_WIDE_CHANNELS = 256


def enabled() -> bool:
    return os.environ.get("FPSG_WINOGRAD", "1") != "0"


def eligible(x: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    def is_(v, want):
        return v == want or v == (want, want)
    return (enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and is_(conv.kernel_size, 3) and is_(conv.stride, 1) and is_(conv.padding, 1) and is_(conv.dilation, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros"
            and min(conv.in_channels, conv.out_channels) >= MIN_CHANNELS
            and max(conv.in_channels, conv.out_channels) >= MIN_WIDE_CHANNELS
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def _call(name, *args):
    _hip.check(getattr(_hip.load(), name)(*args), name)


def _filter(w, flip):
    K, C = w.shape[0], w.shape[1]
    U = torch.empty((16, C, K) if flip else (16, K, C), dtype=torch.float32, device=w.device)
    _call("fpsg_wino_filter_transform", _hip.ptr(w), K, C, 1 if flip else 0, _hip.ptr(U), _hip.stream_of(w))
    return U


def _input(x):
    N, C, H, W = x.shape
    V = torch.empty((16, C, N * (H // 2) * (W // 2)), dtype=torch.float32, device=x.device)
    _call("fpsg_wino_input_transform", _hip.ptr(x), N, C, H, W, _hip.ptr(V), _hip.stream_of(x))
    return V


def _output(M, N, H, W):
    K = M.shape[1]
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=M.device)
    _call("fpsg_wino_output_transform", _hip.ptr(M), N, K, H, W, _hip.ptr(y), _hip.stream_of(M))
    return y


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x, w = x.contiguous(), w.contiguous()
        N, C, H, W = x.shape
        with torch.cuda.device(x.device):
            V = _input(x)
            y = _output(torch.bmm(_filter(w, False), V), N, H, W)
        ctx.save_for_backward(V if ctx.needs_input_grad[1] else None, w)
        ctx.dims = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        V, w = ctx.saved_tensors
        N, C, H, W = ctx.dims
        K = w.shape[0]
        gy = gy.contiguous()
        gx = gw = None
        with torch.cuda.device(gy.device):
            if ctx.needs_input_grad[0]:
                gx = _output(torch.bmm(_filter(w, True), _input(gy)), N, H, W)
            if ctx.needs_input_grad[1]:
                dM = torch.empty((16, K, V.shape[2]), dtype=torch.float32, device=gy.device)
                _call("fpsg_wino_grad_output_transform", _hip.ptr(gy), N, K, H, W, _hip.ptr(dM), _hip.stream_of(gy))
                dU = torch.bmm(dM, V.transpose(1, 2))
                gw = torch.empty_like(w)
                _call("fpsg_wino_filter_grad_transform", _hip.ptr(dU), K, C, _hip.ptr(gw), _hip.stream_of(gy))
        return gx, gw


def conv3x3(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``F.conv2d(x, weight, None, 1, 1)`` for ``x [N,C,H,W]`` (H, W even), ``weight [K,C,3,3]``."""
    if x.dim() != 4 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] != x.shape[1]:
        raise ValueError(f"conv3x3: x {tuple(x.shape)} / weight {tuple(weight.shape)}")
    if x.shape[2] % 2 or x.shape[3] % 2:
        raise ValueError("conv3x3: H and W must be even")
    return _Conv3x3.apply(x, weight)
