"""3x3 convolution through Winograd F(m x m, 3x3), m = 2 or 4 (K6), for the deep layers of the image trunk.

``conv3x3(x, weight)`` is ``F.conv2d(x, weight, None, stride=1, padding=1)`` with gradients to
both arguments.  The 16 transform-domain GEMMs are fp32 batched matrix products on the MFMA
pipes (``torch.bmm`` = hipBLASLt); the input / output / filter transforms are the HIP kernels
of ``csrc/winograd.hip`` through the C ABI.  2.25x (m=2) / 4x (m=4) fewer multiplications than the
direct form; ``tile_size`` picks m = 4 where the image is a multiple of 4 and at least 28 wide
(the 112x112, 56x56 and 28x28 stages) and, with half-empty edge tiles, where its sides are even and at least 12
(the 14x14 stage as a 4x4 grid of tiles: ``ragged_enabled``), m = 2 otherwise; ``FPSG_WINOGRAD_M=2`` forces m = 2.
fp32 error against a float64 convolution: ~1e-6 of the output scale for m = 2, ~1e-5 for m = 4
(the library's own kernels: ~1e-6), inside the 1e-4 budget of the path.  Measured on MI355X against MIOpen's own fp32 Winograd kernel (the solver it picks for these
layers) this is faster from 64 channels up with 4x4 tiles, where the 4x larger transform-domain tensors are
small next to the GEMM work (``profiles/``: wino_*).  Reference layers: the Conv2d(3x3, pad 1) of
torchvision's ``vgg16_bn.features`` built at ``src/models/image_net.py:14``.

The forward keeps the transformed input ``V`` (not ``x``) for the weight gradient
``dU = dM V^T``; with 288 GB of HBM the 4x larger tensor (<= 475 MB per layer here) is cheap.
"""
from __future__ import annotations

import os

import torch

from . import _hip, gemm_split
from .bn_counters import count_batch

# Below these widths the transforms' HBM traffic (4x / 2.25x the image tensors) outweighs the saved
# multiplications (profiles/: wino_bench): 4x4 tiles pay off from 64 channels (conv1_2 ... conv4_3 of
# VGG16), 2x2 tiles (the 14x14 stage) from min(C,K) >= 128 and max(C,K) >= 256; conv1_1 (3 input
# channels) stays with the library.
MIN_CHANNELS_M4 = 64
MIN_CHANNELS_M2 = 128
MIN_WIDE_CHANNELS_M2 = 256


def enabled() -> bool:
    return os.environ.get("FPSG_WINOGRAD", "1") != "0"


def _eligible_layer(conv: torch.nn.Conv2d, H: int, W: int) -> bool:
    def is_(v, want):
        return v == want or v == (want, want)
    return (enabled() and is_(conv.kernel_size, 3) and is_(conv.stride, 1) and is_(conv.padding, 1) and is_(conv.dilation, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and H % 2 == 0 and W % 2 == 0
            and _wide_enough(conv.in_channels, conv.out_channels, tile_size(H, W)))


def eligible(x: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _eligible_layer(conv, x.shape[2], x.shape[3]))


def ragged_enabled() -> bool:
    """F(4x4,3x3) on images whose (even) sides are not multiples of 4 -- VGG's 14 x 14 stage as a 4 x 4 grid of tiles
    whose last row and column are half empty: 16 tiles per image and 36 products instead of 49 tiles and 16 products,
    1.36x fewer multiplications and smaller transform-domain tensors (the three 512-channel layers' products
    145 -> 110 us each at 37 images).  ``FPSG_WINOGRAD_RAGGED=0``: F(2x2) there, as before (A/B)."""
    return os.environ.get("FPSG_WINOGRAD_RAGGED", "1") != "0"


def tile_size(H: int, W: int) -> int:
    forced = os.environ.get("FPSG_WINOGRAD_M")
    if forced:
        return int(forced)
    if H % 4 == 0 and W % 4 == 0 and min(H, W) >= 28:
        return 4
    if ragged_enabled() and H % 2 == 0 and W % 2 == 0 and min(H, W) >= 12 and (H % 4 or W % 4):
        return 4                      # ceil(H/4) x ceil(W/4) tiles; from 12 up the half-empty edge tiles cost less than F(2x2)'s extra products
    return 2


def _tiles(H: int, W: int, m: int) -> int:
    return -(-H // m) * -(-W // m)


def fused_enabled() -> bool:
    """K6f (one kernel per 64-input-channel convolution instead of transform + GEMM + transform):
    at 37 images 0.66 ms against 0.98 ms for the forward of conv1_2 (same for its data gradient);
    it does not leave the transformed input behind, which the weight gradient then rebuilds
    (0.3 ms) -- net +2.3 % episodes/s on c5.  ``FPSG_WINOGRAD_FUSED=0`` switches it off."""
    return os.environ.get("FPSG_WINOGRAD_FUSED", "1") != "0"


def _can_fuse(m: int, c_in: int, c_out: int, n_pixels: int, H: int = 4, W: int = 4) -> bool:
    """``n_pixels`` = N*H*W of the tensor the kernel reads: K6f addresses it with 32-bit byte offsets (< 4 GiB).  K6f tiles
    whole 4x4 output blocks: planes whose sides are not multiples of 4 (F(4x4) with ragged edge tiles, e.g. 14x14) stay
    on the three-kernel form."""
    return (fused_enabled() and m == 4 and c_in == 64 and c_out % 16 == 0 and n_pixels * 64 * 4 < (1 << 32)
            and H % 4 == 0 and W % 4 == 0)


MIN_DW_TILES = 16384      # below (2 images at 224x224: 6272 tiles) the per-range partial sums cost more than K6w saves


def _can_fuse_dw(m: int, c_in: int, c_out: int, N: int, H: int, W: int) -> bool:
    """K6w (``fpsg_wino_dw_fused``): the weight gradient of a 64-input-channel layer in one pass -- no transformed
    input, no transformed output gradient, no GEMM with the tile-long reduction; taken from ``MIN_DW_TILES`` tiles
    up (1-shot episodes stay with the three-kernel form: 256 vs 255 episodes/s).  ``FPSG_WINOGRAD_DW=0``: the
    three-kernel form (A/B)."""
    return (fused_enabled() and os.environ.get("FPSG_WINOGRAD_DW", "1") != "0" and m == 4 and c_in == 64
            and c_out % 16 == 0 and H % 4 == 0 and W % 16 == 0 and N * max(c_in, c_out) * H * W * 4 < (1 << 31)
            and N * (H // 4) * (W // 4) >= MIN_DW_TILES)


def _fused_dw(x, chan, pre_bias, gy):
    """dU [36, K, 64] of the convolution whose input was ``x`` (or relu(BN(x)) with ``chan``) and output gradient ``gy``."""
    N, C, H, W = x.shape
    K = gy.shape[1]
    lib = _hip.load()
    dU = torch.empty((36, K, C), dtype=torch.float32, device=x.device)
    ws = torch.empty((lib.fpsg_wino_dw_fused_workspace_floats(N, K, H, W),), dtype=torch.float32, device=x.device)
    opt = lambda t: _hip.ptr(t) if t is not None else None
    _call("fpsg_wino_dw_fused", _hip.ptr(x), opt(chan), opt(pre_bias), _hip.ptr(gy), N, C, K, H, W, _hip.ptr(dU),
          _hip.ptr(ws), _hip.stream_of(x))
    return dU


def _wide_enough(c_in: int, c_out: int, m: int) -> bool:
    lo, hi = min(c_in, c_out), max(c_in, c_out)
    return lo >= MIN_CHANNELS_M4 if m == 4 else (lo >= MIN_CHANNELS_M2 and hi >= MIN_WIDE_CHANNELS_M2)


def _call(name, *args):
    _hip.check(getattr(_hip.load(), name)(*args), name)


def _bmm(A, B):
    """The transform-domain products ``M[xi] = A[xi] . B[xi]`` (forward: U . V, data gradient: U' . V'): the library's
    fp32-MFMA batched GEMM, or -- ``FPSG_GEMM_SPLIT=1``, layers of at least 128 channels -- K10 on the bf16 matrix pipe
    with exactly split operands (``fpsg_amd/gemm_split.py``: fp32-grade, not the library's bits; opt-in)."""
    if gemm_split.enabled() and min(A.shape[1], A.shape[2]) >= 128 and A.is_cuda:
        return gemm_split.bmm_split(A.contiguous(), B.contiguous(), False)
    return torch.bmm(A, B)


def _bmm_nt(A, B):
    """The weight gradient's ``dU[xi] = dM[xi] . V[xi]^T`` (reduction over the tiles, contiguous in both operands)."""
    if gemm_split.enabled() and min(A.shape[1], B.shape[1]) >= 128 and A.is_cuda:
        return gemm_split.bmm_split(A.contiguous(), B.contiguous(), True)
    return torch.bmm(A, B.transpose(1, 2))


_frozen_cache = None     # {(data_ptr, m, flip): U} while a ``weights_frozen`` block is active


class _FilterBank:
    """The transformed filters of the layers a training step uses, in persistent buffers, all refreshed by ONE launch
    at the start of a ``weights_frozen`` block (``fpsg_wino_filter_transform_batch``) instead of one launch per layer
    and direction when first needed -- 24 launches of a few microseconds for the VGG trunk, which is what a one-episode
    step (configs[1], hipGraph) notices.  An entry is registered the first time a filter is asked for inside a block
    (outside a capture); it holds an alias of the weight, so the memory it reads stays allocated.  Buffers live across
    steps: a step's backward has finished with them (same stream) before the next block's refresh overwrites them.
    A hipGraph captured inside a block holds no filter launches and reads the bank's buffers -- every replay runs
    inside a block that refreshed them -- so an entry a capture has seen is pinned for the life of the process; the
    others are dropped when a whole block goes by without asking for them (another model's weights, a parameter that
    was moved into a flat buffer)."""

    MAX_ENTRIES = 96

    def __init__(self):
        self.entries = {}        # (data_ptr, m, flip) -> [weight alias, U, K, C, used in this block, pinned]
        self.table = None        # [(device, int64 [n, 6] of the registered jobs there, workgroups)]
        self.valid = False       # refreshed in the current outermost block

    def register(self, key, w, U):
        if len(self.entries) < self.MAX_ENTRIES:
            self.entries[key] = [w, U, w.shape[0], w.shape[1], True, False]
            self.table = None

    def lookup(self, key, w, capturing):
        ent = self.entries.get(key) if self.valid else None
        if ent is None or ent[2] != w.shape[0] or ent[3] != w.shape[1] or ent[1].device != w.device:
            return None
        ent[4] = True
        ent[5] = ent[5] or capturing
        return ent[1]

    def has_pinned(self) -> bool:
        return any(e[5] for e in self.entries.values())

    def refresh(self):
        """Transforms every registered filter into its buffer: one launch per device that holds entries (one device
        per process in this framework; a second device gets its own table rather than a stale buffer -- a captured
        hipGraph reads the pinned buffers, so none of them may be left behind)."""
        self.valid = False
        stale = [k for k, e in self.entries.items() if not (e[4] or e[5])]
        for k in stale:
            del self.entries[k]
        if stale:
            self.table = None
        if not self.entries:
            return
        if self.table is None:
            per_dev = {}
            for (ptr, m, flip), (w, U, K, C, _, _) in self.entries.items():
                rows, first = per_dev.setdefault(U.device, [[], 0])
                rows.append([ptr, U.data_ptr(), K, C, 2 * m + (1 if flip else 0), first])
                per_dev[U.device][1] = first + (K * C + 255) // 256
            self.table = [(dev, torch.tensor(rows, dtype=torch.int64).to(dev), blocks)
                          for dev, (rows, blocks) in per_dev.items()]
        for e in self.entries.values():
            e[4] = False
        for dev, table, blocks in self.table:
            with torch.cuda.device(dev):
                _call("fpsg_wino_filter_transform_batch", _hip.ptr(table), table.shape[0], blocks,
                      torch.cuda.current_stream(dev).cuda_stream)
        self.valid = True


_bank = _FilterBank()


def check_bank_before_replay() -> None:
    """A hipGraph captured inside a ``weights_frozen`` block holds no filter-transform launches: it reads the bank's
    pinned buffers, which are only right when the block the replay runs in has refreshed them.  Raises otherwise (a
    replay outside a block would silently use the previous step's filters)."""
    if _bank.has_pinned() and not _bank.valid:
        raise RuntimeError("hipGraph replay outside a refreshed winograd.weights_frozen() block: the captured "
                           "episode reads transformed filters that were not recomputed for the current weights")


def filter_bank_enabled() -> bool:
    """``FPSG_FILTER_BANK=0``: every filter transform in its own launch, when first needed (A/B measurements)."""
    return os.environ.get("FPSG_FILTER_BANK", "1") != "0"


class _ReadOnlyView:
    """The block's cache as a capture sees it when the block is ``constant``: entries made before the capture can be
    read (ordinary persistent tensors that outlive every replay inside the block), nothing is added (a capture's own
    tensors live in the graph's pool)."""

    def __init__(self, d):
        self._d = d

    def __contains__(self, k):
        return k in self._d

    def __getitem__(self, k):
        return self._d[k]

    def get(self, k, default=None):
        return self._d.get(k, default)

    def __setitem__(self, k, v):
        pass

    def __iter__(self):
        return iter(self._d)


_frozen_constant = False     # the active block promised that weights AND running statistics stay constant for its whole life


class weights_frozen:
    """Context manager: the convolution weights do not change inside the block (the episodes of
    one optimizer step), so each transformed filter is computed once and reused: the filters registered by earlier
    steps all at once when the block opens (``_FilterBank``), others when first needed.  The step-scoped cache is not
    used while a hipGraph is being captured (a replay must not reuse that step's tensors); the bank is.
    ``constant=True`` (the evaluation loop: nothing changes for the block's whole life, and graphs captured inside are
    replayed only inside): a capture may READ the entries made before it."""

    def __init__(self, constant: bool = False):
        self._constant = bool(constant)

    def __enter__(self):
        global _frozen_cache, _frozen_constant
        self._outer = _frozen_cache
        self._outer_constant = _frozen_constant
        _frozen_constant = self._constant or (_frozen_constant and _frozen_cache is not None)
        _frozen_cache = {} if _frozen_cache is None else _frozen_cache
        # (a graph captured with the bank reads its pinned buffers whatever FPSG_FILTER_BANK says later)
        if self._outer is None and (filter_bank_enabled() or _bank.has_pinned()) and torch.cuda.is_available() \
                and not torch.cuda.is_current_stream_capturing():
            _bank.refresh()
        return self

    def __exit__(self, *exc):
        global _frozen_cache, _frozen_constant
        _frozen_cache = self._outer
        _frozen_constant = self._outer_constant
        if self._outer is None:
            _bank.valid = False
        return False


def frozen_cache():
    """The step-scoped cache of a ``weights_frozen`` block (None outside one, or while a hipGraph is captured):
    derived forms of the weights -- transformed filters here, the decoder's stacked weights -- keyed by the caller."""
    if _frozen_cache is None:
        return None
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return _ReadOnlyView(_frozen_cache) if _frozen_constant else None
    return _frozen_cache


def frozen_cache_ro():
    """The step-scoped cache for READING, also while a hipGraph is captured (entries made before the capture are
    ordinary persistent tensors; a capture must only never ADD its own)."""
    return _frozen_cache


def eval_chan(gamma, beta, running_mean, running_var, eps, C, dev):
    """Evaluation-mode BatchNorm inside a ``weights_frozen`` block (the evaluation loop, ``evaluate_Network.py:107-118``:
    neither the affine parameters nor the running statistics change): the channel coefficients ``chan [4, C]`` (scale,
    shift, mean, rstd) are computed ONCE per BatchNorm and block -- ``fpsg_bn_stats(training = 0)``, one launch -- and
    every later call of the layer passes them in (``training = 2`` of the K5 forward entry points) instead of launching
    the coefficient kernel again: 23 launches per evaluated item of the PointNet model.  -> ``(chan, mode)``; outside a
    block (or for a BatchNorm first seen while a graph is captured) a fresh buffer and mode 0, as before."""
    cache = _frozen_cache
    if cache is None or running_mean is None or running_var is None or os.environ.get("FPSG_EVAL_CHAN_CACHE", "1") == "0":
        return torch.empty((4, C), dtype=torch.float32, device=dev), 0
    key = ("bn_chan", running_mean.data_ptr(), running_var.data_ptr(), gamma.data_ptr() if gamma is not None else 0,
           beta.data_ptr() if beta is not None else 0, float(eps))
    chan = cache.get(key)
    if chan is not None:
        return chan, 2
    chan = torch.empty((4, C), dtype=torch.float32, device=dev)
    if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return chan, 0
    opt = lambda t: _hip.ptr(t) if t is not None else None
    with torch.cuda.device(dev):
        _call("fpsg_bn_stats", _hip.ptr(chan), None, opt(gamma), opt(beta), _hip.ptr(running_mean), _hip.ptr(running_var),
              0.0, 1, C, 1, 0, float(eps), _hip.ptr(chan), None, None, None, None, 0,
              torch.cuda.current_stream(dev).cuda_stream)
    cache[key] = chan
    return chan, 2


def _filter(m, w, flip):
    key = (w.data_ptr(), m, bool(flip))
    capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()
    banked = _bank.lookup(key, w, capturing)
    if banked is not None:
        return banked
    cache = _frozen_cache
    if cache is not None and capturing:
        cache = _ReadOnlyView(cache) if _frozen_constant else None
    if cache is not None and key in cache:
        return cache[key]
    K, C = w.shape[0], w.shape[1]
    a2 = (m + 2) ** 2
    U = torch.empty((a2, C, K) if flip else (a2, K, C), dtype=torch.float32, device=w.device)
    _call("fpsg_wino_filter_transform", m, _hip.ptr(w), K, C, 1 if flip else 0, _hip.ptr(U), _hip.stream_of(w))
    if cache is not None and not capturing:
        cache[key] = U
        if filter_bank_enabled():
            _bank.register(key, w.detach(), U)      # refreshed with the others from the next block on
    return U


def row_stride(P: int) -> int:
    """The row stride (floats) of the transform-domain tensors ``[A*A, channels, stride]`` for ``P`` tiles: ``P`` rounded
    up to ``FPSG_WINO_ROW_ALIGN`` floats (default 32 = one 128-byte line; 1 = dense rows as before round 5).  The
    trunk's P (29008 / 7252 / 1813 / 592) is never a whole number of lines, and the library's batched products run
    4-13 % faster on line-aligned rows (``profiles/r05/wino_row_stride.txt``).  The pad columns are zeros (written by
    the transforms), the products simply run over ``stride`` columns, the output transforms read the first ``P``."""
    a = int(os.environ.get("FPSG_WINO_ROW_ALIGN", "32"))
    if a < 1 or a > 1024:
        raise ValueError(f"FPSG_WINO_ROW_ALIGN must be in 1 .. 1024 (got {a})")
    return (P + a - 1) // a * a


def _ldp(T, N, H, W, m):
    """The row stride of a transform-domain tensor handed to an output transform: its own last dimension (a dense
    ``[A*A, K, P]`` tensor built by a caller or a padded one from ``_input`` / the products)."""
    if not T.is_contiguous() or T.shape[2] < N * _tiles(H, W, m):
        raise ValueError(f"transform-domain tensor {tuple(T.shape)} must be contiguous with at least "
                         f"{N * _tiles(H, W, m)} columns")
    return T.shape[2]


def _input(m, x):
    N, C, H, W = x.shape
    Ps = row_stride(N * _tiles(H, W, m))
    V = torch.empty(((m + 2) ** 2, C, Ps), dtype=torch.float32, device=x.device)
    _call("fpsg_wino_input_transform", m, _hip.ptr(x), N, C, H, W, _hip.ptr(V), Ps, _hip.stream_of(x))
    return V


def stats_enabled() -> bool:
    """The output transform also accumulates the statistics of the BatchNorm that follows the convolution
    (K5's statistics pass then does not read the tensor again).  ``FPSG_CONV_STATS=0`` switches it off (A/B)."""
    return os.environ.get("FPSG_CONV_STATS", "1") != "0"


_BN_SMALL_MAX = 16384     # fpsg_bn_act_bwd does tensors of at most this many values per channel in one fused kernel


def bwd_stats_enabled() -> bool:
    """The output transform of a data-gradient convolution also accumulates the two sums the backward of the
    BatchNorm + ReLU in front of that convolution starts from (K5's backward then does not read the two tensors for
    them).  ``FPSG_CONV_BWD_STATS=0`` switches it off (A/B)."""
    return os.environ.get("FPSG_CONV_BWD_STATS", "1") != "0"


def _output_bwd_stats(m, M, N, H, W, xpre, pre_bias, chan):
    """-> (ga, parts [K, S, 2]): the output transform of ``M`` and, per channel and workgroup, sum(dz) and
    sum(dz * xhat) for dz = ga * [bn(xpre + pre_bias) > 0] (``fpsg_wino_output_transform_bwd_stats``)."""
    K = M.shape[1]
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=M.device)
    S = _hip.load().fpsg_wino_stats_parts(m, N, H, W)
    parts = torch.empty((K, S, 2), dtype=torch.float32, device=M.device)
    _call("fpsg_wino_output_transform_bwd_stats", m, _hip.ptr(M), N, K, H, W, _hip.ptr(y), _hip.ptr(xpre),
          _hip.ptr(pre_bias) if pre_bias is not None else None, _hip.ptr(chan), _hip.ptr(parts), _ldp(M, N, H, W, m),
          _hip.stream_of(M))
    return y, parts


def _output(m, M, N, H, W, stats_bias=None, want_parts=False):
    """``want_parts``: -> (y, parts [K, S, 2]) with the per-workgroup partial sums of ``y + stats_bias`` and its square."""
    K = M.shape[1]
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=M.device)
    if not want_parts:
        _call("fpsg_wino_output_transform", m, _hip.ptr(M), N, K, H, W, _hip.ptr(y), _ldp(M, N, H, W, m), _hip.stream_of(M))
        return y
    S = _hip.load().fpsg_wino_stats_parts(m, N, H, W)
    parts = torch.empty((K, S, 2), dtype=torch.float32, device=M.device)
    _call("fpsg_wino_output_transform_stats", m, _hip.ptr(M), N, K, H, W, _hip.ptr(y),
          _hip.ptr(stats_bias) if stats_bias is not None else None, _hip.ptr(parts), _ldp(M, N, H, W, m), _hip.stream_of(M))
    return y, parts


def _fused_stats(x, chan, pre_bias, U, stats_bias):
    """K6f (plain: ``chan`` None; or applying BatchNorm + ReLU while loading) -> (y, parts [K, S, 2])."""
    N, C, H, W = x.shape
    K = U.shape[1]
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
    S = _hip.load().fpsg_wino_conv_fused_parts(N, K, H, W)
    parts = torch.empty((K, S, 2), dtype=torch.float32, device=x.device)
    opt = lambda t: _hip.ptr(t) if t is not None else None
    _call("fpsg_wino_conv_fused_stats", _hip.ptr(x), opt(chan), opt(pre_bias), _hip.ptr(U), N, C, K, H, W, _hip.ptr(y),
          opt(stats_bias), _hip.ptr(parts), _hip.stream_of(x))
    return y, parts


def _fused(x, U):
    """y = conv(x, w) for 64 input channels, one kernel (K6f); U = _filter(4, w, flip)."""
    N, C, H, W = x.shape
    K = U.shape[1]
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
    _call("fpsg_wino_conv_fused", _hip.ptr(x), _hip.ptr(U), N, C, K, H, W, _hip.ptr(y), _hip.stream_of(x))
    return y


def _grad_output(m, gy):
    N, K, H, W = gy.shape
    Ps = row_stride(N * _tiles(H, W, m))
    dM = torch.empty(((m + 2) ** 2, K, Ps), dtype=torch.float32, device=gy.device)
    _call("fpsg_wino_grad_output_transform", m, _hip.ptr(gy), N, K, H, W, _hip.ptr(dM), Ps, _hip.stream_of(gy))
    return dM


def grad_transforms_enabled() -> bool:
    """``FPSG_GRAD_TRANSFORMS=0``: the output gradient's two transforms as two launches (A/B measurements)."""
    return os.environ.get("FPSG_GRAD_TRANSFORMS", "1") != "0"


def _grad_transforms(m, gy):
    """``(_input(m, gy), _grad_output(m, gy))`` from ONE pass over ``gy`` (``fpsg_wino_grad_transforms``): the weight
    gradient's tile is the interior of the data gradient's patch.  Bit-identical to the two launches."""
    N, K, H, W = gy.shape
    Ps = row_stride(N * _tiles(H, W, m))
    V = torch.empty(((m + 2) ** 2, K, Ps), dtype=torch.float32, device=gy.device)
    dM = torch.empty_like(V)
    _call("fpsg_wino_grad_transforms", m, _hip.ptr(gy), N, K, H, W, _hip.ptr(V), _hip.ptr(dM), Ps, _hip.stream_of(gy))
    return V, dM


def _filter_grad(m, dU, like):
    gw = torch.empty_like(like)
    _call("fpsg_wino_filter_grad_transform", m, _hip.ptr(dU), like.shape[0], like.shape[1], _hip.ptr(gw),
          _hip.stream_of(dU))
    return gw


class _Conv3x3(torch.autograd.Function):
    """``want_parts``: the op returns ``(y, parts)``; ``parts`` (not differentiable) are the
    statistics partial sums of ``y + stats_bias`` for the BatchNorm that follows (``_output``)."""

    @staticmethod
    def forward(ctx, x, w, m, stats_bias=None, want_parts=False):
        ctx.set_materialize_grads(False)        # no zero tensor for the (non-differentiable) parts output
        x, w = x.contiguous(), w.contiguous()
        N, C, H, W = x.shape
        K = w.shape[0]
        parts = None
        with torch.cuda.device(x.device):
            if _can_fuse(m, C, K, N * H * W, H, W):
                if want_parts:
                    y, parts = _fused_stats(x, None, None, _filter(m, w, False), stats_bias)
                else:
                    y = _fused(x, _filter(m, w, False))
                keep, kept_is_v = x, False                 # V is rebuilt for the weight gradient
            else:
                V = _input(m, x)
                Mt = _bmm(_filter(m, w, False), V)
                if want_parts:
                    y, parts = _output(m, Mt, N, H, W, stats_bias, True)
                else:
                    y = _output(m, Mt, N, H, W)
                keep, kept_is_v = V, True
        ctx.save_for_backward(keep if ctx.needs_input_grad[1] else None, w)
        ctx.dims = (N, C, H, W, m, kept_is_v)
        if not want_parts:
            return y
        if parts is not None:
            ctx.mark_non_differentiable(parts)
        return y, parts

    @staticmethod
    def backward(ctx, gy, _gparts=None):
        if gy is None:
            return None, None, None, None, None
        kept, w = ctx.saved_tensors
        N, C, H, W, m, kept_is_v = ctx.dims
        K = w.shape[0]
        gy = gy.contiguous()
        gx = gw = None
        with torch.cuda.device(gy.device):
            fuse_dx = _can_fuse(m, K, C, N * H * W, H, W)                   # the data gradient is a convolution K -> C
            fuse_dw = not kept_is_v and _can_fuse_dw(m, C, K, N, H, W)
            Vg = dM = None
            if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not fuse_dx and not fuse_dw
                    and grad_transforms_enabled()):
                Vg, dM = _grad_transforms(m, gy)                            # both transforms of gy from one read
            if ctx.needs_input_grad[0]:
                if fuse_dx:
                    gx = _fused(gy, _filter(m, w, True))
                else:
                    gx = _output(m, _bmm(_filter(m, w, True), Vg if Vg is not None else _input(m, gy)), N, H, W)
                    Vg = None
            if ctx.needs_input_grad[1]:
                if fuse_dw:
                    gw = _filter_grad(m, _fused_dw(kept, None, None, gy), w)
                else:
                    V = kept if kept_is_v else _input(m, kept)
                    gw = _filter_grad(m, _bmm_nt(dM if dM is not None else _grad_output(m, gy), V), w)
        return gx, gw, None, None, None


def _input_act(m, x, chan, pre_bias):
    N, C, H, W = x.shape
    Ps = row_stride(N * _tiles(H, W, m))
    V = torch.empty(((m + 2) ** 2, C, Ps), dtype=torch.float32, device=x.device)
    _call("fpsg_wino_input_transform_act", m, _hip.ptr(x), _hip.ptr(chan), _hip.ptr(pre_bias) if pre_bias is not None else None,
          N, C, H, W, _hip.ptr(V), Ps, _hip.stream_of(x))
    return V


def _fused_act(x, chan, pre_bias, U):
    N, C, H, W = x.shape
    K = U.shape[1]
    y = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
    _call("fpsg_wino_conv_fused_act", _hip.ptr(x), _hip.ptr(chan), _hip.ptr(pre_bias) if pre_bias is not None else None,
          _hip.ptr(U), N, C, K, H, W, _hip.ptr(y), _hip.stream_of(x))
    return y


def fold_enabled() -> bool:
    """``FPSG_BN_FOLD=0``: BatchNorm + ReLU between two Winograd convolutions as its own K5 pass (A/B)."""
    return os.environ.get("FPSG_BN_FOLD", "1") != "0"


class _BNReluConv3x3(torch.autograd.Function):
    """``conv3x3(relu(BN(y + pre_bias)), w)`` for the PRE-BatchNorm output ``y`` of the previous convolution,
    without the activation tensor: K5's statistics pass (``fpsg_bn_stats``), then the Winograd input
    transform (or K6f) applies scale / shift / ReLU while loading.  The backward is the convolution's
    (data + weight gradient, ``V`` kept or rebuilt through the same activating transform) followed by K5's
    BatchNorm + ReLU backward on ``y``.  Forward and backward values are those of ``bn_act`` + ``conv3x3``
    bit for bit (same kernels' arithmetic, one pass less over the tensor)."""

    @staticmethod
    def forward(ctx, y, pre_bias, gamma, beta, running_mean, running_var, training, momentum, eps, w, m, parts=None,
                stats_bias=None, want_parts=False):
        """``parts``: statistics partial sums of ``y + pre_bias`` from the convolution that produced ``y`` (the
        pass over ``y`` is skipped); ``stats_bias`` / ``want_parts``: the same for this convolution's output
        (-> ``(out, out_parts)``)."""
        ctx.set_materialize_grads(False)        # no zero tensor for the (non-differentiable) parts output
        y, w = y.contiguous(), w.contiguous()
        N, C, H, W = y.shape
        K = w.shape[0]
        lib = _hip.load()
        dev = y.device
        chan, mode = (torch.empty((4, C), dtype=torch.float32, device=dev), 1) if training else \
            eval_chan(gamma, beta, running_mean, running_var, eps, C, dev)
        use_parts = parts is not None and training
        ws = None if (use_parts or mode == 2) else torch.empty((lib.fpsg_bn_workspace_floats(N, C, H * W),), dtype=torch.float32, device=dev)
        opt = lambda t: _hip.ptr(t) if t is not None else None
        out_parts = None
        with torch.cuda.device(dev):
            if mode != 2:       # (evaluation inside a weights_frozen block: the coefficients are the block's cached ones)
                _call("fpsg_bn_stats", _hip.ptr(y), opt(pre_bias), opt(gamma), opt(beta), opt(running_mean), opt(running_var),
                      float(momentum), N, C, H * W, 1 if training else 0, float(eps), _hip.ptr(chan), None, None,
                      opt(ws), _hip.ptr(parts) if use_parts else None, parts.shape[1] if use_parts else 0,
                      _hip.stream_of(y))
            if _can_fuse(m, C, K, N * H * W, H, W):
                if want_parts:
                    out, out_parts = _fused_stats(y, chan, pre_bias, _filter(m, w, False), stats_bias)
                else:
                    out = _fused_act(y, chan, pre_bias, _filter(m, w, False))
                V = None                                     # rebuilt for the weight gradient
            else:
                V = _input_act(m, y, chan, pre_bias)
                Mt = _bmm(_filter(m, w, False), V)
                if want_parts:
                    out, out_parts = _output(m, Mt, N, H, W, stats_bias, True)
                else:
                    out = _output(m, Mt, N, H, W)
        ctx.save_for_backward(y, chan, pre_bias, w, V if ctx.needs_input_grad[9] else None)
        ctx.cfg = (N, C, H, W, m, bool(training), gamma is not None, beta is not None)
        if not want_parts:
            return out
        if out_parts is not None:
            ctx.mark_non_differentiable(out_parts)
        return out, out_parts

    @staticmethod
    def backward(ctx, gout, _gparts=None):
        if gout is None:
            return (None,) * 14
        y, chan, pre_bias, w, V = ctx.saved_tensors
        N, C, H, W, m, training, has_g, has_b = ctx.cfg
        K = w.shape[0]
        lib = _hip.load()
        dev = y.device
        gout = gout.contiguous()
        gw = None
        with torch.cuda.device(dev):
            # convolution backward: gradient of the (never stored) activation, and of the filter
            bwd_parts = None
            fuse_dx = _can_fuse(m, K, C, N * H * W, H, W)
            fuse_dw = V is None and _can_fuse_dw(m, C, K, N, H, W)
            Vg = dM = None
            if ctx.needs_input_grad[9] and not fuse_dx and not fuse_dw and grad_transforms_enabled():
                Vg, dM = _grad_transforms(m, gout)                          # both transforms of gout from one read
            if fuse_dx:
                ga = _fused(gout, _filter(m, w, True))
            elif bwd_stats_enabled() and N * H * W > _BN_SMALL_MAX:
                # the output transform that writes ga also delivers the sums K5's backward starts from
                ga, bwd_parts = _output_bwd_stats(m, _bmm(_filter(m, w, True), Vg if Vg is not None else _input(m, gout)),
                                                  N, H, W, y, pre_bias, chan)
            else:
                ga = _output(m, _bmm(_filter(m, w, True), Vg if Vg is not None else _input(m, gout)), N, H, W)
            Vg = None
            if ctx.needs_input_grad[9]:
                if fuse_dw:
                    gw = _filter_grad(m, _fused_dw(y, chan, pre_bias, gout), w)
                else:
                    if V is None:
                        V = _input_act(m, y, chan, pre_bias)
                    gw = _filter_grad(m, _bmm_nt(dM if dM is not None else _grad_output(m, gout), V), w)
            # BatchNorm + ReLU backward on y (K5)
            want_dpb = pre_bias is not None and ctx.needs_input_grad[1]
            dy = torch.empty_like(y)
            dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
            dpb = torch.empty((C,), dtype=torch.float32, device=dev) if want_dpb else None
            coef = torch.empty((3, C), dtype=torch.float32, device=dev)
            ws = torch.empty((lib.fpsg_bn_workspace_floats(N, C, H * W),), dtype=torch.float32, device=dev)
            if bwd_parts is not None:
                _call("fpsg_bn_act_bwd_parts", _hip.ptr(y), _hip.ptr(pre_bias) if pre_bias is not None else None,
                      _hip.ptr(ga), _hip.ptr(chan), N, C, H * W, 1 if training else 0, 1, 0.0, _hip.ptr(dy),
                      _hip.ptr(dgamma), _hip.ptr(dbeta), _hip.ptr(dpb) if want_dpb else None, _hip.ptr(coef),
                      _hip.ptr(ws), _hip.ptr(bwd_parts), bwd_parts.shape[1], _hip.stream_of(y))
            else:
                _call("fpsg_bn_act_bwd", _hip.ptr(y), _hip.ptr(pre_bias) if pre_bias is not None else None, _hip.ptr(ga),
                      _hip.ptr(chan), N, C, H * W, 1 if training else 0, 1, 0.0, _hip.ptr(dy), _hip.ptr(dgamma),
                      _hip.ptr(dbeta), _hip.ptr(dpb) if want_dpb else None, _hip.ptr(coef), _hip.ptr(ws), _hip.stream_of(y))
        return (dy, dpb, dgamma if has_g else None, dbeta if has_b else None, None, None, None, None, None, gw, None,
                None, None, None)


def stem_fold_enabled() -> bool:
    """``FPSG_STEM_FOLD=0``: conv1_1 and the fused BatchNorm + ReLU + conv1_2 stay two autograd functions, the
    BatchNorm's dx pass writes dy for K8 (A/B measurements)."""
    return os.environ.get("FPSG_STEM_FOLD", "1") != "0"


class _StemConvBNReluConv(torch.autograd.Function):
    """The trunk's first two convolutions as ONE autograd function (training mode):
    ``conv3x3(relu(BN(conv3x3_first(x, w1) + b1)), w2)`` = ``vgg16_bn.features[0:4]`` without conv1_2's bias
    (reference ``src/models/image_net.py:14``).  Forward: K8f, K5's statistics from its epilogue sums, the activating
    K6f / K6 -- the kernels ``conv3x3_first`` + ``_BNReluConv3x3`` launch.  What the fusion buys is the BACKWARD: the
    gradient of conv1_1's output is consumed by exactly one kernel, K8's weight gradient (images carry no gradient), so
    the BatchNorm backward stops after its sums (``fpsg_bn_act_bwd_coef``) and K8 forms ``dy`` tile by tile from ``y1``,
    the activation gradient and the coefficients (``fpsg_conv_first_dw_fold``): the 475 MB ``dy`` of a 37-image episode
    is neither written nor read.  Same values as the two functions, bit for bit, except the gradient of conv1_1's bias:
    in front of a training-mode BatchNorm it is zero in exact arithmetic (sum(dx) = (k1 - scale) sum(dz), k1 = scale);
    the two-function form returns K5's round-off for it, this one returns 0."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, running_mean, running_var, momentum, eps, w2, m, stats_bias=None,
                want_parts=False):
        from . import conv_first
        ctx.set_materialize_grads(False)
        y1, parts1 = conv_first._forward(x, w1, b1, True)
        N, C, H, W = y1.shape
        K = w2.shape[0]
        lib = _hip.load()
        dev = y1.device
        chan = torch.empty((4, C), dtype=torch.float32, device=dev)
        opt = lambda t: _hip.ptr(t) if t is not None else None
        out_parts = None
        w2c = w2.contiguous()
        with torch.cuda.device(dev):
            _call("fpsg_bn_stats", _hip.ptr(y1), opt(b1), opt(gamma), opt(beta), opt(running_mean), opt(running_var),
                  float(momentum), N, C, H * W, 1, float(eps), _hip.ptr(chan), None, None, None, _hip.ptr(parts1),
                  parts1.shape[1], _hip.stream_of(y1))
            if _can_fuse(m, C, K, N * H * W, H, W):
                if want_parts:
                    out, out_parts = _fused_stats(y1, chan, b1, _filter(m, w2c, False), stats_bias)
                else:
                    out = _fused_act(y1, chan, b1, _filter(m, w2c, False))
                V = None
            else:
                V = _input_act(m, y1, chan, b1)
                Mt = _bmm(_filter(m, w2c, False), V)
                if want_parts:
                    out, out_parts = _output(m, Mt, N, H, W, stats_bias, True)
                else:
                    out = _output(m, Mt, N, H, W)
        ctx.save_for_backward(x, w1, y1, chan, b1, w2c, V if ctx.needs_input_grad[9] else None)
        ctx.cfg = (N, C, H, W, m, gamma is not None, beta is not None)
        if not want_parts:
            return out
        if out_parts is not None:
            ctx.mark_non_differentiable(out_parts)
        return out, out_parts

    @staticmethod
    def backward(ctx, gout, _gparts=None):
        if gout is None:
            return (None,) * 13
        x, w1, y1, chan, b1, w2, V = ctx.saved_tensors
        N, C, H, W, m, has_g, has_b = ctx.cfg
        K = w2.shape[0]
        lib = _hip.load()
        dev = y1.device
        gout = gout.contiguous()
        gw1 = gw2 = db1 = None
        with torch.cuda.device(dev):
            if _can_fuse(m, K, C, N * H * W, H, W):
                ga = _fused(gout, _filter(m, w2, True))
            else:
                ga = _output(m, _bmm(_filter(m, w2, True), _input(m, gout)), N, H, W)
            if ctx.needs_input_grad[9]:
                if V is None and _can_fuse_dw(m, C, K, N, H, W):
                    gw2 = _filter_grad(m, _fused_dw(y1, chan, b1, gout), w2)
                else:
                    if V is None:
                        V = _input_act(m, y1, chan, b1)
                    gw2 = _filter_grad(m, _bmm_nt(_grad_output(m, gout), V), w2)
            dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
            coef = torch.empty((3, C), dtype=torch.float32, device=dev)
            ws = torch.empty((lib.fpsg_bn_workspace_floats(N, C, H * W),), dtype=torch.float32, device=dev)
            _call("fpsg_bn_act_bwd_coef", _hip.ptr(y1), _hip.ptr(b1), _hip.ptr(ga), _hip.ptr(chan), N, C, H * W, 1, 1, 0.0,
                  _hip.ptr(dgamma), _hip.ptr(dbeta), _hip.ptr(coef), _hip.ptr(ws), _hip.stream_of(y1))
            if ctx.needs_input_grad[1]:
                gw1 = torch.empty_like(w1, memory_format=torch.contiguous_format)
                ws1 = torch.empty((lib.fpsg_conv_first_dw_workspace_floats(N, H, W),), dtype=torch.float32, device=dev)
                xc = x.contiguous()
                _call("fpsg_conv_first_dw_fold", _hip.ptr(xc), _hip.ptr(y1), _hip.ptr(ga), _hip.ptr(chan), _hip.ptr(coef),
                      _hip.ptr(b1), N, 3, 64, H, W, _hip.ptr(gw1), _hip.ptr(ws1), _hip.stream_of(y1))
            if ctx.needs_input_grad[2]:
                db1 = torch.zeros((C,), dtype=torch.float32, device=dev)
        return (None, gw1, db1, dgamma if has_g else None, dbeta if has_b else None, None, None, None, None, gw2, None,
                None, None)


def stem_applies(x, conv1, bn1, conv2) -> bool:
    """``stem_conv_bn_relu_conv`` serves: training mode, the 3 -> 64 first layer on K8f, images without a gradient, planes
    beyond K5's small-tensor limit, a Winograd-eligible 64-channel second layer."""
    from . import conv_first
    if not (stem_fold_enabled() and fold_enabled() and bn1.training and bn1.track_running_stats and stats_enabled()):
        return False
    if x.requires_grad or not conv_first.eligible(x, conv1) or os.environ.get("FPSG_CONV_FIRST_FWD", "1") == "0":
        return False
    if conv1.bias is None or conv2.bias is None or x.shape[0] * x.shape[2] * x.shape[3] <= _BN_SMALL_MAX:
        return False
    return (isinstance(conv2, torch.nn.Conv2d) and conv2.in_channels == 64
            and _eligible_layer(conv2, x.shape[2], x.shape[3]))


def stem_conv_bn_relu_conv(x, conv1, bn1, conv2, want_parts=False):
    """``conv2`` (bias-free) of ``relu(bn1(conv1(x)))`` -> the pre-BatchNorm output of conv1_2 (and, with ``want_parts``,
    the statistics partial sums of ``y + conv2.bias``); updates ``bn1``'s running statistics as the module would."""
    count_batch(bn1)
    mom = 0.1 if bn1.momentum is None else float(bn1.momentum)
    m = tile_size(x.shape[2], x.shape[3])
    return _StemConvBNReluConv.apply(x.contiguous(), conv1.weight, conv1.bias, bn1.weight, bn1.bias, bn1.running_mean,
                                     bn1.running_var, mom, bn1.eps, conv2.weight, m, conv2.bias, bool(want_parts))


def bn_relu_conv3x3(y, pre_bias, bn, weight, m=None, parts=None, stats_bias=None, want_parts=False):
    """``conv3x3(relu(bn(y + pre_bias)), weight)`` with ``bn`` an ``nn.BatchNorm2d`` in its current mode; updates
    its running statistics / ``num_batches_tracked`` as the module would.  ``parts`` / ``stats_bias`` /
    ``want_parts``: see ``_BNReluConv3x3.forward``."""
    if m is None:
        m = tile_size(y.shape[2], y.shape[3])
    training = bn.training
    if training:
        count_batch(bn)
    mom = (0.1 if bn.momentum is None else float(bn.momentum)) if training else -1.0
    return _BNReluConv3x3.apply(y, pre_bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(training), mom,
                                bn.eps, weight, m, parts, stats_bias, bool(want_parts))


def conv3x3(x: torch.Tensor, weight: torch.Tensor, m: int | None = None, stats_bias=None, want_parts: bool = False):
    """``F.conv2d(x, weight, None, 1, 1)`` for ``x [N,C,H,W]`` (H, W even), ``weight [K,C,3,3]``;
    ``m``: output tile size 2 or 4 (default: ``tile_size(H, W)``).  ``want_parts``: -> ``(y, parts)``, the statistics
    partial sums of ``y + stats_bias`` for a following BatchNorm (None where the kernel does not deliver them)."""
    if x.dim() != 4 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] != x.shape[1]:
        raise ValueError(f"conv3x3: x {tuple(x.shape)} / weight {tuple(weight.shape)}")
    if m is None:
        m = tile_size(x.shape[2], x.shape[3])
    if m not in (2, 4) or x.shape[2] % 2 or x.shape[3] % 2:
        raise ValueError(f"conv3x3: m={m} must be 2 or 4 and H, W even (got {tuple(x.shape[2:])})")
    if want_parts:
        return _Conv3x3.apply(x, weight, m, stats_bias, True)
    return _Conv3x3.apply(x, weight, m)
