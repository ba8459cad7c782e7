"""The first convolution of the VGG16-BN trunk (3 -> 64 channels, 3x3, padding 1;
``vgg16_bn.features[0]``, reference ``src/models/image_net.py:14``).

The forward is K8f (``fpsg_conv_first_fwd``: bound by the write of the output; optionally also the partial sums of
the BatchNorm statistics that follow), the WEIGHT gradient K8 (``fpsg_conv_first_dw``): ``dy`` -- 475 MB at 37
images -- is read once, where the library spends two layout transposes and an NHWC implicit GEMM on a
1,728-element result.  The data gradient, which the train step never asks for (images carry no gradient), is
the library's.  ``FPSG_CONV_FIRST_FWD=0``: the library's forward (A/B)."""
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


def _forward(x, w, stats_bias=None, want_parts=False):
    xc, wc = x.contiguous(), w.contiguous()
    N, _, H, W = xc.shape
    lib = _hip.load()
    y = torch.empty((N, 64, H, W), dtype=torch.float32, device=x.device)
    parts = None
    if want_parts:
        parts = torch.empty((64, lib.fpsg_conv_first_parts(N, H, W), 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.fpsg_conv_first_fwd(_hip.ptr(xc), _hip.ptr(wc), N, 3, 64, H, W, _hip.ptr(y),
                                     _hip.ptr(stats_bias) if (want_parts and stats_bias is not None) else None,
                                     _hip.ptr(parts) if want_parts else None, _hip.stream_of(xc))
    _hip.check(rc, "fpsg_conv_first_fwd")
    return y, parts


class _ConvFirst(torch.autograd.Function):
    """``want_parts``: -> ``(y, parts)`` with the statistics partial sums of ``y + stats_bias`` (not differentiable)."""

    @staticmethod
    def forward(ctx, x, w, stats_bias=None, want_parts=False):
        ctx.set_materialize_grads(False)        # no zero tensor for the (non-differentiable) parts output
        ctx.save_for_backward(x, w)
        if os.environ.get("FPSG_CONV_FIRST_FWD", "1") == "0":
            y, parts = F.conv2d(x, w, None, 1, 1), None
        else:
            y, parts = _forward(x, w, stats_bias, want_parts)
        if not want_parts:
            return y
        if parts is not None:
            ctx.mark_non_differentiable(parts)
        return y, parts

    @staticmethod
    def backward(ctx, gy, _gparts=None):
        if gy is None:
            return None, None, None, None
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
        return gx, gw, None, None


def conv3x3_first(x: torch.Tensor, weight: torch.Tensor, stats_bias=None, want_parts: bool = False):
    """``F.conv2d(x, weight, None, 1, 1)`` for ``x [N,3,H,W]``, ``weight [64,3,3,3]``; ``want_parts``: ->
    ``(y, parts)``, the statistics partial sums of ``y + stats_bias`` for a following BatchNorm."""
    if want_parts:
        return _ConvFirst.apply(x, weight, stats_bias, True)
    return _ConvFirst.apply(x, weight)
