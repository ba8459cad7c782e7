"""BatchNorm fused with its activation (K5) for the module pairs of the reference networks.

``bn_act(bn, x, act)`` is ``act(bn(x))`` for an ``nn.BatchNorm1d/2d`` module ``bn`` and
``act`` in {None, 'relu', ('leaky', slope)}: same training / eval semantics, same running
statistics update, same gradients.  On a ROCm tensor with at least 64 values per (sample,
channel) row it is ONE fused HIP op (2 reads + 1 write forward, 4 reads + 1 write backward,
only ``x`` saved); otherwise (CPU tensors of the CPU port, ``nn.Linear`` outputs) the plain
PyTorch modules run -- that is device dispatch of host plumbing, not a fallback for a missing
library: on a GPU tensor a missing ``libfpsg_hip.so`` raises.

``conv_bn_act(conv, bn, x, act)`` is ``act(bn(conv(x)))`` for the Conv + BatchNorm + activation
triples: on the fused path the convolution runs WITHOUT its bias, which K5 adds in registers
(``pre_bias``), and the bias gradient comes out of K5's dx pass -- the library path spends a
broadcast-add kernel (read + write of the activation tensor) and a full reduction of dx on it.

``FPSG_FUSED_BN=0`` switches the fused op off (A/B measurements).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .bn_counters import count_batch
from . import _hip, conv_first, winograd

_ACT_CODES = {None: 0, "none": 0, "relu": 1, "leaky": 2}
_MIN_ROW = 64
ROWS_SEGMENT_MAX = 16384       # bnact.hip's kBnSmallMax: the longest column segment of fpsg_bn_act_rows_*


def fused_enabled() -> bool:
    return os.environ.get("FPSG_FUSED_BN", "1") != "0"


def _parse_act(act):
    if isinstance(act, tuple):
        return _ACT_CODES[act[0]], float(act[1])
    return _ACT_CODES[act], 0.0


def _chan_and_mode(training, weight, bias, running_mean, running_var, eps, C, dev):
    """``(chan [4, C], training argument of the K5 forward)``: 1 = batch statistics; 0 = evaluation, coefficients formed by
    the call; 2 = evaluation inside a ``winograd.weights_frozen`` block, coefficients formed once for the block
    (``winograd.eval_chan``).  ``FPSG_EVAL_CHAN_CACHE=0``: always 0 in evaluation (A/B)."""
    if training:
        return torch.empty((4, C), dtype=torch.float32, device=dev), 1
    if os.environ.get("FPSG_EVAL_CHAN_CACHE", "1") == "0":
        return torch.empty((4, C), dtype=torch.float32, device=dev), 0
    return winograd.eval_chan(weight, bias, running_mean, running_var, eps, C, dev)


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, eps, act_code, slope, want_stats,
                pre_bias=None, momentum=-1.0):
        N, C = x.shape[0], x.shape[1]
        L = x.numel() // (N * C)
        lib = _hip.load()
        dev = x.device
        ctx.set_materialize_grads(False)        # no zero tensors for the (non-differentiable) statistics outputs
        y = torch.empty_like(x)
        chan, mode = _chan_and_mode(training, weight, bias, running_mean, running_var, eps, C, dev)
        ws = torch.empty((lib.fpsg_bn_workspace_floats(N, C, L),), dtype=torch.float32, device=dev)
        bmean = bvar = None
        if training and want_stats:
            # the two rows of one tensor: callers that post-process both (point_cloud_net._update_running) do it in one op
            bmean, bvar = torch.empty((2, C), dtype=torch.float32, device=dev).unbind(0)
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_fwd(
                _hip.ptr(x), _hip.ptr(pre_bias) if pre_bias is not None else None,
                _hip.ptr(weight) if weight is not None else None,
                _hip.ptr(bias) if bias is not None else None,
                _hip.ptr(running_mean) if running_mean is not None else None,
                _hip.ptr(running_var) if running_var is not None else None, float(momentum),
                N, C, L, mode, float(eps), act_code, float(slope), _hip.ptr(y), _hip.ptr(chan),
                _hip.ptr(bmean) if bmean is not None else None, _hip.ptr(bvar) if bvar is not None else None,
                _hip.ptr(ws), _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_fwd")
        ctx.save_for_backward(x, chan, pre_bias)
        ctx.cfg = (N, C, L, training, act_code, slope, weight is not None, bias is not None)
        ctx.mark_non_differentiable(*(t for t in (bmean, bvar) if t is not None))
        if training and want_stats:
            return y, bmean, bvar
        return y, None, None

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        if gy is None:
            return (None,) * 12
        x, chan, pre_bias = ctx.saved_tensors
        N, C, L, training, act_code, slope, has_w, has_b = ctx.cfg
        want_dpb = pre_bias is not None and ctx.needs_input_grad[10]
        lib = _hip.load()
        dev = x.device
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        dpb = torch.empty((C,), dtype=torch.float32, device=dev) if want_dpb else None
        coef = torch.empty((3, C), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.fpsg_bn_workspace_floats(N, C, L),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_bwd(_hip.ptr(x), _hip.ptr(pre_bias) if pre_bias is not None else None,
                                     _hip.ptr(gy), _hip.ptr(chan), N, C, L, 1 if training else 0,
                                     act_code, float(slope), _hip.ptr(dx), _hip.ptr(dgamma), _hip.ptr(dbeta),
                                     _hip.ptr(dpb) if want_dpb else None, _hip.ptr(coef), _hip.ptr(ws),
                                     _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_bwd")
        return (dx, (dgamma if has_w else None), (dbeta if has_b else None), None, None, None, None, None, None,
                None, dpb, None)


class _BNActRows(torch.autograd.Function):
    """Training-mode ``act(BN(x + pre_bias))`` over column segments of the rows of ``x [..., M]`` (contiguous): segment
    ``i`` = the next ``seg_lens[i]`` columns is one BatchNorm call with its own statistics
    (``fpsg_bn_act_rows_fwd/bwd``: one launch for all segments).  Returns ``(y, stats [nseg, 2, rows])`` -- batch mean
    and unbiased batch variance per segment and row; the affine parameters' gradients are summed over the segments."""

    @staticmethod
    def forward(ctx, x, weight, bias, pre_bias, seg_lens, eps, act_code, slope):
        import ctypes
        M = x.shape[-1]
        rows = x.numel() // M
        seg_lens = [int(n) for n in seg_lens]
        if sum(seg_lens) != M or not x.is_contiguous():
            raise ValueError(f"_BNActRows: segments {seg_lens} must tile the contiguous rows of {tuple(x.shape)}")
        nseg = len(seg_lens)
        offs = [sum(seg_lens[:i]) for i in range(nseg)]
        c_off, c_len = (ctypes.c_int * nseg)(*offs), (ctypes.c_int * nseg)(*seg_lens)
        lib = _hip.load()
        dev = x.device
        ctx.set_materialize_grads(False)
        y = torch.empty_like(x)
        chan = torch.empty((nseg, 4, rows), dtype=torch.float32, device=dev)
        stats = torch.empty((nseg, 2, rows), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_rows_fwd(_hip.ptr(x), M, c_off, c_len, nseg,
                                          _hip.ptr(pre_bias) if pre_bias is not None else None, _hip.ptr(weight),
                                          _hip.ptr(bias), rows, float(eps), act_code, float(slope), _hip.ptr(y),
                                          _hip.ptr(chan), _hip.ptr(stats), _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_rows_fwd")
        ctx.save_for_backward(x, chan, pre_bias)
        ctx.cfg = (M, rows, offs, seg_lens, act_code, slope)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gstats):
        if gy is None:
            return (None,) * 8
        import ctypes
        x, chan, pre_bias = ctx.saved_tensors
        M, rows, offs, seg_lens, act_code, slope = ctx.cfg
        nseg = len(seg_lens)
        c_off, c_len = (ctypes.c_int * nseg)(*offs), (ctypes.c_int * nseg)(*seg_lens)
        want_dpb = pre_bias is not None and ctx.needs_input_grad[3]
        dev = x.device
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        grads = torch.empty((3 if want_dpb else 2, nseg, rows), dtype=torch.float32, device=dev)   # dgamma, dbeta, dpre_bias
        with torch.cuda.device(dev):
            rc = _hip.load().fpsg_bn_act_rows_bwd(
                _hip.ptr(x), M, c_off, c_len, nseg, _hip.ptr(pre_bias) if pre_bias is not None else None, _hip.ptr(gy),
                _hip.ptr(chan), rows, act_code, float(slope), _hip.ptr(dx), _hip.ptr(grads[0]), _hip.ptr(grads[1]),
                _hip.ptr(grads[2]) if want_dpb else None, _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_rows_bwd")
        g = grads.sum(1) if nseg > 1 else grads[:, 0]          # the segments share the affine parameters and the bias
        return dx, g[0], g[1], (g[2] if want_dpb else None), None, None, None, None


class _BNActPool(torch.autograd.Function):
    """``max_pool2d(act(BN(x + pre_bias)), 2)`` as one op (K5 pooled variants): the forward writes
    only the pooled tensor, the backward re-derives the windows' arg-max from ``x``."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, eps, act_code, slope, pre_bias,
                momentum=-1.0, parts=None):
        """``parts``: statistics partial sums of ``x + pre_bias`` delivered by the producing convolution."""
        N, C, H, W = x.shape
        lib = _hip.load()
        dev = x.device
        yp = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=dev)
        chan, mode = _chan_and_mode(training, weight, bias, running_mean, running_var, eps, C, dev)
        use_parts = parts is not None and training
        ws = None if use_parts else torch.empty((lib.fpsg_bn_pool_workspace_floats(N, C, H, W),), dtype=torch.float32,
                                                device=dev)
        opt = lambda t: _hip.ptr(t) if t is not None else None
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_pool_fwd(_hip.ptr(x), opt(pre_bias), opt(weight), opt(bias), opt(running_mean),
                                          opt(running_var), float(momentum), N, C, H, W, mode,
                                          float(eps), act_code, float(slope), _hip.ptr(yp), _hip.ptr(chan), None, None,
                                          opt(ws), _hip.ptr(parts) if use_parts else None,
                                          parts.shape[1] if use_parts else 0, _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_pool_fwd")
        ctx.save_for_backward(x, chan, pre_bias)
        ctx.cfg = (N, C, H, W, training, act_code, slope, weight is not None, bias is not None)
        return yp

    @staticmethod
    def backward(ctx, gyp):
        x, chan, pre_bias = ctx.saved_tensors
        N, C, H, W, training, act_code, slope, has_w, has_b = ctx.cfg
        want_dpb = pre_bias is not None and ctx.needs_input_grad[9]
        lib = _hip.load()
        dev = x.device
        gyp = gyp.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        dpb = torch.empty((C,), dtype=torch.float32, device=dev) if want_dpb else None
        coef = torch.empty((3, C), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.fpsg_bn_pool_workspace_floats(N, C, H, W),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_pool_bwd(_hip.ptr(x), _hip.ptr(pre_bias) if pre_bias is not None else None,
                                          _hip.ptr(gyp), _hip.ptr(chan), N, C, H, W, 1 if training else 0, act_code,
                                          float(slope), _hip.ptr(dx), _hip.ptr(dgamma), _hip.ptr(dbeta),
                                          _hip.ptr(dpb) if want_dpb else None, _hip.ptr(coef), _hip.ptr(ws),
                                          _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_pool_bwd")
        return (dx, (dgamma if has_w else None), (dbeta if has_b else None), None, None, None, None, None, None, dpb, None,
                None)


class _BNActMax(torch.autograd.Function):
    """``act(BN(x + pre_bias)).max(dim=2)[0]`` for ``x [N, C, L]`` as one op (K5 max variants): only
    each row's extreme of ``x`` is normalised; the backward scatters the ``[N, C]`` gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, eps, act_code, slope, pre_bias,
                momentum=-1.0):
        N, C, L = x.shape
        lib = _hip.load()
        dev = x.device
        out = torch.empty((N, C), dtype=torch.float32, device=dev)
        idx = torch.empty((N, C), dtype=torch.int32, device=dev)
        chan, mode = _chan_and_mode(training, weight, bias, running_mean, running_var, eps, C, dev)
        ws = torch.empty((lib.fpsg_bn_max_workspace_floats(N, C, L),), dtype=torch.float32, device=dev)
        opt = lambda t: _hip.ptr(t) if t is not None else None
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_max_fwd(_hip.ptr(x), opt(pre_bias), opt(weight), opt(bias), opt(running_mean),
                                         opt(running_var), float(momentum), N, C, L, mode, float(eps),
                                         act_code, float(slope), _hip.ptr(out), _hip.ptr(idx), _hip.ptr(chan), None,
                                         None, _hip.ptr(ws), _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_max_fwd")
        ctx.save_for_backward(x, chan, idx, pre_bias)
        ctx.cfg = (N, C, L, training, act_code, slope, weight is not None, bias is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, chan, idx, pre_bias = ctx.saved_tensors
        N, C, L, training, act_code, slope, has_w, has_b = ctx.cfg
        want_dpb = pre_bias is not None and ctx.needs_input_grad[9]
        lib = _hip.load()
        dev = x.device
        gout = gout.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        dpb = torch.empty((C,), dtype=torch.float32, device=dev) if want_dpb else None
        coef = torch.empty((3, C), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.fpsg_bn_max_workspace_floats(N, C, L),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_max_bwd(_hip.ptr(x), _hip.ptr(pre_bias) if pre_bias is not None else None,
                                         _hip.ptr(gout), _hip.ptr(idx), _hip.ptr(chan), N, C, L,
                                         1 if training else 0, act_code, float(slope), _hip.ptr(dx),
                                         _hip.ptr(dgamma), _hip.ptr(dbeta), _hip.ptr(dpb) if want_dpb else None,
                                         _hip.ptr(coef), _hip.ptr(ws), _hip.stream_of(x))
        _hip.check(rc, "fpsg_bn_act_max_bwd")
        return dx, (dgamma if has_w else None), (dbeta if has_b else None), None, None, None, None, None, None, dpb, None


def W_is_plain(conv) -> bool:
    return conv.weight.is_contiguous() and conv.weight.dim() == 3 and conv.weight.shape[2] == 1


# Below this many points over the batch (8 clouds of 2048) the dense gradient's two GEMMs are small and the algebraic
# form's dozen launches cost more than they save (1-shot episodes: 2 clouds; 287 vs 262 episodes/s).
_MAX_ALGEBRA_MIN_POINTS = 16384


def max_bwd_algebra_enabled() -> bool:
    """The backward of conv1x1 -> BatchNorm (+ReLU) -> max over the points through K x K algebra instead of two GEMMs
    over the dense [B,C,L] gradient (``_ConvBNActMax``).  ``FPSG_MAX_BWD_ALGEBRA=0``: the dense form (A/B)."""
    return os.environ.get("FPSG_MAX_BWD_ALGEBRA", "1") != "0"


class _ConvBNActMax(torch.autograd.Function):
    """``act(BN(W a + conv_bias)).max(dim=2)[0]`` for ``a [B, K, L]``, ``weight [C, K, 1]`` as ONE op: the tail of
    PointNet's shared MLPs (``pointnet/model.py:35-37, 222-224``).  Forward = the batched product + ``_BNActMax``'s
    kernels.  Backward: the gradient of the pre-BatchNorm tensor is ``k1 dz [l = sel] + k2 x' + k3`` (sparse + affine
    in ``x' = W a + b``), so the two products over it (``dW = dx' a^T``, ``da = W^T dx'``: 34 GFLOP each at B = 64,
    C = 1024, L = 2048, plus a 537 MB write and two reads of ``dx'``) reduce to the K x K Gram matrix ``G = a a^T``, K x K
    products, one [K x K].[K x B L] GEMM and a gather / scatter of B*C columns (``csrc/maxbwd.hip``): 8.6 GFLOP.
    Same mathematics, sums reassociated (1e-6 of the gradients' scale); deterministic."""

    @staticmethod
    def forward(ctx, a, weight, conv_bias, gamma, beta, running_mean, running_var, training, eps, act_code, slope, momentum):
        B, K, L = a.shape
        C = weight.shape[0]
        lib = _hip.load()
        dev = a.device
        W = weight.reshape(C, K)
        x = torch.bmm(W.unsqueeze(0).expand(B, -1, -1), a)              # [B, C, L], bias-free
        out = torch.empty((B, C), dtype=torch.float32, device=dev)
        idx = torch.empty((B, C), dtype=torch.int32, device=dev)
        chan, mode = _chan_and_mode(training, gamma, beta, running_mean, running_var, eps, C, dev)
        ws = torch.empty((lib.fpsg_bn_max_workspace_floats(B, C, L),), dtype=torch.float32, device=dev)
        opt = lambda t: _hip.ptr(t) if t is not None else None
        with torch.cuda.device(dev):
            rc = lib.fpsg_bn_act_max_fwd(_hip.ptr(x), opt(conv_bias), opt(gamma), opt(beta), opt(running_mean),
                                         opt(running_var), float(momentum), B, C, L, mode, float(eps),
                                         act_code, float(slope), _hip.ptr(out), _hip.ptr(idx), _hip.ptr(chan), None,
                                         None, _hip.ptr(ws), _hip.stream_of(a))
        _hip.check(rc, "fpsg_bn_act_max_fwd")
        ctx.save_for_backward(a, weight, x, chan, idx, conv_bias)
        ctx.cfg = (B, K, C, L, bool(training), act_code, slope, gamma is not None, beta is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, weight, x, chan, idx, pb = ctx.saved_tensors
        B, K, C, L, training, act_code, slope, has_g, has_b = ctx.cfg
        lib = _hip.load()
        dev = a.device
        gout = gout.contiguous()
        W = weight.reshape(C, K)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
        coef = torch.empty((3, C), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.fpsg_bn_max_workspace_floats(B, C, L),), dtype=torch.float32, device=dev)
        st = _hip.stream_of(a)
        with torch.cuda.device(dev):
            _hip.check(lib.fpsg_bn_act_max_bwd_coef(_hip.ptr(x), _hip.ptr(pb) if pb is not None else None, _hip.ptr(gout),
                                                    _hip.ptr(idx), _hip.ptr(chan), B, C, L, 1 if training else 0, act_code,
                                                    float(slope), _hip.ptr(dgamma), _hip.ptr(dbeta), _hip.ptr(coef),
                                                    _hip.ptr(ws), st), "fpsg_bn_act_max_bwd_coef")
            off = lib.fpsg_bn_max_dz_offset(B, C, L)        # the workspace layout is bnact.hip's
            dz = ws[off: off + B * C].view(B, C)
            opt = lambda t: _hip.ptr(t) if t is not None else None
            da = dw = dpb = None
            want_dpb = pb is not None and ctx.needs_input_grad[2]
            # per-channel scalings of the K x K products: one launch (fpsg_max_bwd_prep)
            Wk = torch.empty((C, K), dtype=torch.float32, device=dev)
            u = torch.empty((C,), dtype=torch.float32, device=dev)
            if want_dpb:
                dpb = torch.empty((C,), dtype=torch.float32, device=dev)
            spart = s = S = None
            if ctx.needs_input_grad[1]:
                S = torch.empty((C, K), dtype=torch.float32, device=dev)
                if training:                                                # the gather pass also leaves the row sums of a
                    spart = torch.empty((B, K), dtype=torch.float32, device=dev)
                    s = torch.empty((K,), dtype=torch.float32, device=dev)
                _hip.check(lib.fpsg_max_bwd_gather(_hip.ptr(a), _hip.ptr(dz), _hip.ptr(idx), B, K, C, L, _hip.ptr(S), opt(spart),
                                                   st), "fpsg_max_bwd_gather")
            _hip.check(lib.fpsg_max_bwd_prep(_hip.ptr(W), _hip.ptr(coef), opt(pb), _hip.ptr(chan[2]), _hip.ptr(dz), opt(spart),
                                             B, K, C, L, _hip.ptr(Wk), _hip.ptr(u), opt(dpb), opt(s), st), "fpsg_max_bwd_prep")
            if ctx.needs_input_grad[1]:
                WG = None
                if training:
                    G = torch.bmm(a, a.transpose(1, 2)).sum(0)              # [K, K] Gram matrix
                    WG = W @ G
                dw = torch.empty((C, K), dtype=torch.float32, device=dev)
                _hip.check(lib.fpsg_max_bwd_dw(_hip.ptr(S), opt(WG), _hip.ptr(coef), opt(pb), opt(s), B, K, C, _hip.ptr(dw),
                                               st), "fpsg_max_bwd_dw")
                dw = dw.reshape(weight.shape)
            if ctx.needs_input_grad[0]:
                if training:
                    M = W.t() @ Wk                                          # [K, K] = W^T diag(k2) W
                    v = torch.mv(W.t(), u)                                  # [K]   = W^T (k2 pb + k3)
                    da = torch.bmm(M.unsqueeze(0).expand(B, -1, -1), a)
                else:
                    v = torch.zeros((K,), dtype=torch.float32, device=dev)
                    da = torch.zeros_like(a)
                sws = torch.empty((lib.fpsg_max_bwd_scatter_workspace_floats(B, C, L),), dtype=torch.float32, device=dev)
                _hip.check(lib.fpsg_max_bwd_scatter(_hip.ptr(da), _hip.ptr(W), _hip.ptr(coef), _hip.ptr(dz), _hip.ptr(idx),
                                                    _hip.ptr(v), B, K, C, L, _hip.ptr(sws), st), "fpsg_max_bwd_scatter")
        return (da, dw, dpb, dgamma if has_g else None, dbeta if has_b else None, None, None, None, None, None, None, None)


def _eligible(x: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and fused_enabled()):
        return False
    row = x.numel() // max(x.shape[0] * x.shape[1], 1)
    return row >= _MIN_ROW and row % 4 == 0 and x.shape[1] <= 65535


def _plain_act(y, act_code, slope):
    if act_code == 1:
        return F.relu(y)
    if act_code == 2:
        return F.leaky_relu(y, slope)
    return y


def batch_norm_act(x, weight, bias, running_mean, running_var, training, momentum, eps, act=None,
                   return_stats: bool = False, pre_bias=None):
    """Functional form: ``act(F.batch_norm(x [+ pre_bias per channel], ...))``.  With
    ``return_stats`` (training only) also returns the batch mean and the UNBIASED batch variance
    instead of touching the running buffers (the caller applies its own update rule)."""
    act_code, slope = _parse_act(act)
    if _eligible(x):
        xc = x.contiguous()
        # training without return_stats: the kernel's finalize step also applies the momentum update
        # of the running statistics (no separate elementwise launches)
        update = training and not return_stats and running_mean is not None
        m = (0.1 if momentum is None else float(momentum)) if update else -1.0
        y, bmean, bvar = _BNAct.apply(xc, weight, bias, running_mean, running_var, bool(training), eps,
                                      act_code, slope, bool(training and return_stats), pre_bias, m)
        return (y, bmean, bvar) if return_stats else y
    if pre_bias is not None:
        x = x + pre_bias.view(1, -1, *([1] * (x.dim() - 2)))
    if return_stats:
        C = x.shape[1]
        mean = torch.zeros(C, dtype=x.dtype, device=x.device)
        var = torch.ones(C, dtype=x.dtype, device=x.device)
        y = F.batch_norm(x, mean, var, weight, bias, True, 1.0, eps)      # mean/var <- batch statistics
        return _plain_act(y, act_code, slope), mean, var
    y = F.batch_norm(x, running_mean, running_var, weight, bias, training, 0.1 if momentum is None else momentum, eps)
    return _plain_act(y, act_code, slope)


def batch_norm_act_rows(x, weight, bias, seg_lens, eps, act=None, pre_bias=None):
    """Training-mode ``act(batch_norm(x + pre_bias))`` with independent batch statistics per column segment of the rows
    of ``x [..., M]`` (K5, one launch): -> ``(y, stats [nseg, 2, rows])``, the batch mean and the unbiased batch variance
    of every segment.  ROCm fp32 contiguous tensors only; the caller applies the running-statistics updates."""
    act_code, slope = _parse_act(act)
    if not _eligible(x):
        raise RuntimeError("batch_norm_act_rows needs the HIP path (an fp32 tensor on the GPU)")
    return _BNActRows.apply(x, weight, bias, pre_bias, tuple(seg_lens), eps, act_code, slope)


def bn_act(bn: nn.modules.batchnorm._BatchNorm, x: torch.Tensor, act=None, pre_bias=None) -> torch.Tensor:
    """``act(bn(x [+ pre_bias]))`` with the module's parameters, buffers, mode and momentum."""
    if not _eligible(x) or not bn.track_running_stats:
        if pre_bias is not None:
            x = x + pre_bias.view(1, -1, *([1] * (x.dim() - 2)))
        return _plain_act(bn(x), *_parse_act(act))
    training = bn.training
    if training:
        count_batch(bn)
    return batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum,
                          bn.eps, act, pre_bias=pre_bias)


def _is_pointwise(conv) -> bool:
    return (isinstance(conv, nn.Conv1d) and conv.kernel_size == (1,) and conv.stride == (1,)
            and conv.padding == (0,) and conv.dilation == (1,) and conv.groups == 1)


def wants_conv_stats(conv, bn, x) -> bool:
    """The convolution's epilogue should deliver the statistics of ``bn`` (training mode, a Winograd layer, planes
    large enough for the consumers that take them: the pooled K5 pass and the fold into the next convolution)."""
    return (bn.training and winograd.stats_enabled() and isinstance(conv, nn.Conv2d) and x.dim() == 4
            and x.shape[0] * x.shape[2] * x.shape[3] > 16384
            and (winograd.eligible(x, conv) or conv_first.eligible(x, conv)))


def _conv_without_bias(conv, x, want_parts: bool = False):
    """The convolution of ``conv`` without its bias: K6 (Winograd transforms + MFMA batched GEMM)
    for the 3x3 layers; a batched GEMM for the 1x1 ``Conv1d`` of the shared MLPs (the library
    convolution wraps the same product in NHWC transposes of ``[B,1024,N]`` tensors for its
    weight gradient, and its GEMM kernel is not covered by the recorded kernel choices); the
    library convolution otherwise.  ``want_parts``: -> ``(y, parts)``, the statistics partial sums of ``y + conv.bias``
    from the convolution's epilogue (None where the kernel does not deliver them)."""
    if want_parts:
        if isinstance(conv, nn.Conv2d) and winograd.eligible(x, conv):
            return winograd.conv3x3(x, conv.weight, stats_bias=conv.bias, want_parts=True)
        if isinstance(conv, nn.Conv2d) and conv_first.eligible(x, conv):
            return conv_first.conv3x3_first(x, conv.weight, stats_bias=conv.bias, want_parts=True)
        return _conv_without_bias(conv, x), None
    if isinstance(conv, nn.Conv2d) and winograd.eligible(x, conv):
        return winograd.conv3x3(x, conv.weight)
    if isinstance(conv, nn.Conv2d) and conv_first.eligible(x, conv):
        return conv_first.conv3x3_first(x, conv.weight)          # library forward, K8 weight gradient
    if _is_pointwise(conv) and x.is_cuda and x.dim() == 3:
        w = conv.weight.squeeze(-1)                               # [K, C]
        return torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)
    return conv._conv_forward(x, conv.weight, None)


def fusable_conv(conv, bn, x) -> bool:
    """The Conv + BatchNorm pair runs bias-free + K5 pre-bias on this input."""
    return (conv.bias is not None and x.is_cuda and x.dtype == torch.float32 and fused_enabled()
            and bn.track_running_stats and conv.padding_mode == "zeros")


def foldable_into(y: torch.Tensor, bn, next_conv) -> bool:
    """BatchNorm(+ReLU) of the bias-free convolution output ``y`` can be applied by ``next_conv``'s Winograd input
    transform instead of a K5 apply pass: large planes (the one-launch small-tensor K5 has no statistics-only
    form), a Winograd-eligible consumer."""
    return (winograd.fold_enabled() and isinstance(next_conv, nn.Conv2d) and _eligible(y) and y.dim() == 4
            and y.shape[0] * y.shape[2] * y.shape[3] > 16384 and bn.track_running_stats
            and next_conv.bias is not None and winograd.eligible(y, next_conv))


def bn_act_of(y: torch.Tensor, conv, bn, act=None) -> torch.Tensor:
    """``act(bn(y + conv.bias))`` for the bias-free output ``y`` of ``conv``."""
    if not _eligible(y):
        return bn_act(bn, y + conv.bias.view(1, -1, *([1] * (y.dim() - 2))), act)
    return bn_act(bn, y, act, pre_bias=conv.bias)


def bn_act_pool_of(y: torch.Tensor, conv, bn, pool, act=None, parts=None) -> torch.Tensor:
    """``pool(act(bn(y + conv.bias)))`` for the bias-free output ``y`` of ``conv`` (K5 pooled variant); ``parts``:
    the statistics partial sums the convolution delivered with ``y``."""
    if not _pool_eligible(y, pool):
        return pool(bn_act_of(y, conv, bn, act))
    training = bn.training
    if training:
        count_batch(bn)
    act_code, slope = _parse_act(act)
    m = (0.1 if bn.momentum is None else float(bn.momentum)) if training else -1.0
    return _BNActPool.apply(y.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(training),
                            bn.eps, act_code, slope, conv.bias, m, parts)


def conv_bn_act(conv: nn.modules.conv._ConvNd, bn: nn.modules.batchnorm._BatchNorm, x: torch.Tensor,
                act=None) -> torch.Tensor:
    """``act(bn(conv(x)))`` for ``nn.Conv1d/2d`` + BatchNorm (+ ReLU / LeakyReLU) triples."""
    if conv.bias is None or not (x.is_cuda and x.dtype == torch.float32 and fused_enabled()
                                 and bn.track_running_stats and conv.padding_mode == "zeros"):
        return bn_act(bn, conv(x), act)
    y = _conv_without_bias(conv, x)                        # bias joins inside K5
    if not _eligible(y):
        return bn_act(bn, y + conv.bias.view(1, -1, *([1] * (y.dim() - 2))), act)
    return bn_act(bn, y, act, pre_bias=conv.bias)


def _pool_eligible(y: torch.Tensor, pool: nn.MaxPool2d) -> bool:
    def two(v):
        return v == 2 or v == (2, 2)
    return (_eligible(y) and y.dim() == 4 and y.shape[2] % 2 == 0 and y.shape[3] % 2 == 0
            and two(pool.kernel_size) and two(pool.stride) and pool.padding in (0, (0, 0))
            and pool.dilation in (1, (1, 1)) and not pool.return_indices
            and y.shape[0] * y.shape[2] * y.shape[3] > 16384)      # small planes: the one-launch K5 + torch's pool


def conv_bn_act_pool(conv: nn.Conv2d, bn: nn.BatchNorm2d, pool: nn.MaxPool2d, x: torch.Tensor, act=None):
    """``pool(act(bn(conv(x))))`` for the groups that end a VGG stage (2x2 / stride-2 max-pool)."""
    fused_conv = (conv.bias is not None and x.is_cuda and x.dtype == torch.float32 and fused_enabled()
                  and bn.track_running_stats and conv.padding_mode == "zeros")
    if not fused_conv:
        return pool(conv_bn_act(conv, bn, x, act))
    y = _conv_without_bias(conv, x)
    if not _pool_eligible(y, pool):
        if not _eligible(y):
            return pool(bn_act(bn, y + conv.bias.view(1, -1, 1, 1), act))
        return pool(bn_act(bn, y, act, pre_bias=conv.bias))
    training = bn.training
    if training:
        count_batch(bn)
    act_code, slope = _parse_act(act)
    m = (0.1 if bn.momentum is None else float(bn.momentum)) if training else -1.0
    return _BNActPool.apply(y.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(training),
                            bn.eps, act_code, slope, conv.bias, m)


def conv_bn_act_max(conv: nn.Conv1d, bn: nn.BatchNorm1d, x: torch.Tensor, act=None) -> torch.Tensor:
    """``act(bn(conv(x))).max(dim=2)[0]`` -> ``[N, C]``: the tail of PointNet's shared MLPs
    (``pointnet/model.py:35-37, 222-224``)."""
    fused = (conv.bias is not None and x.is_cuda and x.dtype == torch.float32 and fused_enabled()
             and bn.track_running_stats and conv.padding_mode == "zeros")
    if (fused and max_bwd_algebra_enabled() and _is_pointwise(conv) and x.dim() == 3 and conv.in_channels <= 128
            and x.shape[2] >= _MIN_ROW and x.shape[2] % 4 == 0 and conv.out_channels <= 65535 and x.is_contiguous()
            and x.shape[2] <= 2048 and x.shape[0] * x.shape[2] >= _MAX_ALGEBRA_MIN_POINTS
            and W_is_plain(conv)):
        training = bn.training
        if training:
            count_batch(bn)
        act_code, slope = _parse_act(act)
        m = (0.1 if bn.momentum is None else float(bn.momentum)) if training else -1.0
        return _ConvBNActMax.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bool(training), bn.eps, act_code, slope, m)
    if fused:
        y = _conv_without_bias(conv, x)
        fused = y.dim() == 3 and _eligible(y)
    if not fused:
        return conv_bn_act(conv, bn, x, act).max(dim=2)[0]
    training = bn.training
    if training:
        count_batch(bn)
    act_code, slope = _parse_act(act)
    m = (0.1 if bn.momentum is None else float(bn.momentum)) if training else -1.0
    return _BNActMax.apply(y.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(training),
                           bn.eps, act_code, slope, conv.bias, m)
