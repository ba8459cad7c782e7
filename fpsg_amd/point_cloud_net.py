"""Point-cloud encoder switch and the multi-patch point-set decoder.

Mirrors reference ``src/models/point_cloud_net.py``: ``PointNetWrapper :11-18``,
``PCEncoder :21-34``, ``MLPDeformer :37-55``, ``PrimitiveNode :57-80``,
``PrimitiveCluster :82-112``, ``PCDecoder :114-132``, ``get_activation :135-145`` -- same
constructor arguments, same module tree, hence the same state-dict keys
(``cluster_pool.<i>.deformer.*``, ``cluster_pool.<i>.node_pool.<j>.*``).

MI355X-first change in the arithmetic (exact up to fp32 re-association):
the reference materialises the 1536-d latent ``x.repeat(1, 1, 128)`` once per cluster and
feeds ``cat(x_rep, patch_pts)`` [B,1539,128] through a 1539x1539 1x1 convolution in each of
the 16 nodes.  The latent columns are constant over a patch's points, so
``conv1(cat(x_rep, p)) = (W[:, :1536] x + b)  (one [B,1536]x[1536,1539] GEMM per node)
                        +  W[:, 1536:] p    (a 3-channel 1x1 convolution)``:
no repeat, no concat, and conv1 shrinks from 2.37 M to ~23 K MAC per point.  Training-mode
BatchNorm still sees the same ``B*128`` pre-activations.  ``PrimitiveNode.forward`` keeps the
reference signature (full ``[B,1539,P]`` input) for drop-in use and for the parity test.

The random 2-D grid of every patch is drawn per forward exactly as in the reference
(``utils.py:51-54``); ``PCDecoder.forward(..., grid=..., generator=...)`` additionally lets a
caller inject or seed it.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import bn_counters
from .fused_bn import batch_norm_act, batch_norm_act_rows, bn_act
from .pointnet import PointNetfeat
from .utils import get_template


def get_activation(argument: str):
    table = {
        "relu": F.relu,
        "sigmoid": torch.sigmoid,
        "softplus": F.softplus,
        "logsigmoid": F.logsigmoid,
        "softsign": F.softsign,
        "tanh": torch.tanh,
    }
    if argument not in table:
        raise ValueError(f"Invalid activation: {argument}")
    return table[argument]


def _bn_then(bn, x, activation):
    """``activation(bn(x))``; BatchNorm + ReLU run as one fused pass (K5) on the GPU."""
    if activation is F.relu:
        return bn_act(bn, x, "relu")
    return activation(bn_act(bn, x, None))


class PointNetWrapper(nn.Module):
    def __init__(self):
        super().__init__()
        self.pointnet_feat_extractor = PointNetfeat()

    def forward(self, x):
        return self.pointnet_feat_extractor(x)[0]


class PCEncoder(nn.Module):
    """``core`` in {'pointnet', 'dgcnn'}; ``forward(x[B,3,N]) -> [B,1024]``."""

    def __init__(self, core: str = "pointnet"):
        super().__init__()
        if core == "pointnet":
            self.pc_encoder = PointNetWrapper()
        elif core == "dgcnn":
            from .dgcnn import DGCNNfeat
            self.pc_encoder = DGCNNfeat()
        else:
            raise NotImplementedError(f"Unsupported Point Cloud Encoder Core: {core}")

    def forward(self, x):
        return self.pc_encoder(x)


class MLPDeformer(nn.Module):
    """Per-cluster patch deformer: ``ori_dim -> 128 -> 128 -> raw_dim`` (tanh output)."""

    def __init__(self, conf):
        super().__init__()
        self.layer_size = 128
        self.input_size = conf.ori_dim
        self.dim_output = conf.raw_dim
        self.conv1 = nn.Conv1d(self.input_size, self.layer_size, 1)
        self.conv2 = nn.Conv1d(self.layer_size, self.layer_size, 1)
        self.conv3 = nn.Conv1d(self.layer_size, self.dim_output, 1)
        self.bn1 = nn.BatchNorm1d(self.layer_size)
        self.bn2 = nn.BatchNorm1d(self.layer_size)
        self.activation = get_activation(conf.activation)

    def forward(self, x):
        x = _bn_then(self.bn1, self.conv1(x), self.activation)
        x = _bn_then(self.bn2, self.conv2(x), self.activation)
        return torch.tanh(self.conv3(x))


class PrimitiveNode(nn.Module):
    """One patch MLP: ``D -> D -> D//2 -> D//4 -> 3`` with ``D = raw_dim + bottleneck``."""

    def __init__(self, conf, input_dim: int):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = 3
        d = input_dim
        self.conv1 = nn.Conv1d(d, d, 1)
        self.conv2 = nn.Conv1d(d, d // 2, 1)
        self.conv3 = nn.Conv1d(d // 2, d // 4, 1)
        self.conv4 = nn.Conv1d(d // 4, self.output_dim, 1)
        self.bn1 = nn.BatchNorm1d(d)
        self.bn2 = nn.BatchNorm1d(d // 2)
        self.bn3 = nn.BatchNorm1d(d // 4)
        self.activation = get_activation(conf.activation)

    def _tail(self, h):
        h = _bn_then(self.bn1, h, self.activation)
        h = _bn_then(self.bn2, self.conv2(h), self.activation)
        h = _bn_then(self.bn3, self.conv3(h), self.activation)
        return torch.tanh(self.conv4(h))

    def forward(self, x):
        """Reference signature: ``x [B, D, P]`` (latent already repeated and concatenated)."""
        return self._tail(self.conv1(x))

    def forward_split(self, latent, pts):
        """``latent [B, D - raw]`` (constant over the patch), ``pts [B, raw, P]``."""
        n_lat = latent.size(1)
        w = self.conv1.weight  # [D, D, 1]
        h_lat = F.linear(latent, w[:, :n_lat, 0], self.conv1.bias)  # [B, D]
        h_pts = F.conv1d(pts, w[:, n_lat:, :])                      # [B, D, P]
        return self._tail(h_pts + h_lat.unsqueeze(2))


class PrimitiveCluster(nn.Module):
    def __init__(self, conf, deformer, ttl_pts: int, nodes: int):
        super().__init__()
        self.conf = conf
        self.deformer = deformer
        self.num_nodes = nodes
        self.pts_per_node = ttl_pts // self.num_nodes
        self.template = [get_template(conf.template_type, device=conf.device)
                         for _ in range(self.num_nodes)]
        self.node_pool = nn.ModuleList([
            PrimitiveNode(conf, conf.raw_dim + conf.bottleneck_size) for _ in range(self.num_nodes)
        ])

    def sample_grids(self, batch: int, device, generator=None):
        return [t.get_random_points(torch.Size((batch, t.dim, self.pts_per_node)), device=device,
                                    generator=generator) for t in self.template]

    def forward(self, x, grids=None, generator=None):
        """``x [B, bottleneck]`` -> ``[B, 3, pts_per_node * num_nodes]``."""
        if grids is None:
            grids = self.sample_grids(x.size(0), x.device, generator)
        # one deformer call per patch, as in the reference: each call normalises with its
        # own batch statistics and updates the running statistics once
        patches = [self.deformer(g) for g in grids]
        outs = [node.forward_split(x, p) for node, p in zip(self.node_pool, patches)]
        return torch.cat(outs, dim=2)


class _SplitLast(torch.autograd.Function):
    """``(w[..., :L], w[..., L:])`` as views; the backward writes the two gradients into ONE new
    tensor.  Autograd's own slicing gives each slice a zero-filled full-size gradient and adds the
    two: for the stacked first-layer weight ``[16, 1539, 1539]`` (152 MB) that is five extra passes."""

    @staticmethod
    def forward(ctx, w, L):
        ctx.L = L
        return w[..., :L], w[..., L:]

    @staticmethod
    def backward(ctx, g_lat, g_pts):
        L = ctx.L
        shape = list(g_lat.shape if g_lat is not None else g_pts.shape)
        shape[-1] = (g_lat.shape[-1] if g_lat is not None else L) + (g_pts.shape[-1] if g_pts is not None else 0)
        ref = g_lat if g_lat is not None else g_pts
        gw = torch.empty(shape, dtype=ref.dtype, device=ref.device)
        if g_lat is not None:
            gw[..., :L].copy_(g_lat)
        else:
            gw[..., :L].zero_()
        if g_pts is not None:
            gw[..., L:].copy_(g_pts)
        else:
            gw[..., L:].zero_()
        return gw, None


_MOMENTUM_WEIGHTS: dict = {}


def _momentum_weights(r, m, device):
    key = (r, float(m), str(device))
    w = _MOMENTUM_WEIGHTS.get(key)
    if w is None:
        w = torch.tensor([m * (1 - m) ** (r - 1 - j) for j in range(r)], dtype=torch.float32, device=device).view(1, r, 1)
        _MOMENTUM_WEIGHTS[key] = w
    return w


def _adjacent_rows(a, b):
    """``a`` and ``b`` as one ``[2, n]`` tensor when they are consecutive contiguous rows of the same storage, else None."""
    n = a.numel()
    if (a.dim() == 1 and b.shape == a.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + n):
        return torch.as_strided(a, (2, n), (n, 1), a.storage_offset())
    return None


def _update_running(bns, r, mean, var):
    """``r`` sequential momentum updates per module of ``bns`` from the per-call statistics ``mean`` /
    ``var`` ``[len(bns)*r*C]`` (var unbiased, as torch), in closed form and as multi-tensor ops:
        run <- (1-m)^r run + m * sum_j (1-m)^(r-1-j) stat_j"""
    C = bns[0].num_features
    m = 0.1 if bns[0].momentum is None else bns[0].momentum
    mean3, var3 = mean.view(len(bns), r, C), var.view(len(bns), r, C)
    alpha = 1.0
    if r == 1:
        new_mean, new_var, alpha = mean3[:, 0], var3[:, 0], m          # run <- (1-m) run + m stat: m applied by the add
    else:
        # one weighted sum over the r calls (the weights live on the device, built once per (r, m): the step stays
        # capturable) instead of 2r - 1 elementwise launches per statistic
        coef = _momentum_weights(r, m, mean.device)
        both = _adjacent_rows(mean, var)
        if both is not None:                                        # K5 delivers them as the rows of one tensor
            new_mean, new_var = (both.view(2, len(bns), r, C) * coef).sum(2).unbind(0)
        else:
            new_mean, new_var = (mean3 * coef).sum(1), (var3 * coef).sum(1)
    # (inside the train step's bn_counters.deferred() block these are batched with the other layers' updates)
    bn_counters.update_running([b.running_mean for b in bns] + [b.running_var for b in bns], (1 - m) ** r,
                               list(new_mean.unbind(0)) + list(new_var.unbind(0)), alpha)
    for b in bns:
        bn_counters.count_batch(b, r)


def _update_running_calls(bns, r, stats):
    """``stats [n, 2, len(bns)*r*C]`` (batch mean, unbiased batch variance): the statistics of ``n`` consecutive forward
    passes, each of which calls module ``i`` of ``bns`` ``r`` times -- ``n * r`` sequential momentum updates per module,
    in closed form (see ``_update_running``)."""
    n = stats.shape[0]
    C, K = bns[0].num_features, len(bns)
    m = 0.1 if bns[0].momentum is None else bns[0].momentum
    runs = [b.running_mean for b in bns] + [b.running_var for b in bns]
    if r == 1:
        for i in range(n):          # views only: the updates are the multi-tensor ops of bn_counters
            both = stats[i].view(2, K, C)
            bn_counters.update_running(runs, 1 - m, list(both[0].unbind(0)) + list(both[1].unbind(0)), m)
    else:
        w = _momentum_weights(n * r, m, stats.device).view(n, 1, 1, r, 1)
        new = (stats.view(n, 2, K, r, C) * w).sum((0, 3))                       # [2, K, C]
        bn_counters.update_running(runs, (1 - m) ** (n * r), list(new[0].unbind(0)) + list(new[1].unbind(0)), 1.0)
    for b in bns:
        bn_counters.count_batch(b, n * r)


def _layer1_fused_ok(x, pts, B, P, act) -> bool:
    """K9 applies: ROCm tensors, ReLU, patch sizes the kernel tiles (P = 4 x a power of two <= 256,
    B*P <= 8192).  ``FPSG_DEC1=0`` selects the library chain (A/B measurements)."""
    import os
    return (x.is_cuda and pts.is_cuda and x.dtype == torch.float32 and act is F.relu
            and os.environ.get("FPSG_DEC1", "1") != "0" and P % 4 == 0 and P <= 256
            and ((P // 4) & (P // 4 - 1)) == 0 and B * P <= 8192)


class _DecoderLayer1(torch.autograd.Function):
    """``relu(BN(conv1(cat(x_rep, pts))))`` of all G patch MLPs at once, first layer split into the latent
    GEMM (library, MFMA) and K9 (``fpsg_dec1_fwd/bwd``): the [G,D,B*P] pre-BatchNorm tensor is never
    stored, forward or backward.  ``w1 [G,D,L+3]`` stacked conv1 weights, ``b1 [G,D,1]`` biases, ``x [B,L]``
    latents, ``pts [G,3,B*P]`` deformed patch points.  Returns ``(out [G,D,B*P], batch_mean, batch_var)``."""

    @staticmethod
    def forward(ctx, w1, b1, x, pts, gamma, beta, running_mean, running_var, training, eps, P):
        from . import _hip
        lib = _hip.load()
        G, D, K = w1.shape
        B, L = x.shape
        BP = B * P
        ctx.set_materialize_grads(False)        # no zero tensors for the (non-differentiable) statistics outputs
        w1 = w1.contiguous()
        pts = pts.contiguous()
        hlat = torch.baddbmm(b1, w1[..., :L], x.t().unsqueeze(0).expand(G, L, B))          # [G,D,B]
        out = torch.empty((G, D, BP), dtype=torch.float32, device=x.device)
        chan = torch.empty((4, G * D), dtype=torch.float32, device=x.device)
        bmean = torch.empty((G * D,), dtype=torch.float32, device=x.device) if training else None
        bvar = torch.empty((G * D,), dtype=torch.float32, device=x.device) if training else None
        opt = lambda t: _hip.ptr(t) if t is not None else None
        with torch.cuda.device(x.device):
            rc = lib.fpsg_dec1_fwd(_hip.ptr(hlat), _hip.ptr(w1), K, L, _hip.ptr(pts), _hip.ptr(gamma), _hip.ptr(beta),
                                   opt(running_mean), opt(running_var), G, D, B, P, 1 if training else 0, float(eps),
                                   _hip.ptr(out), _hip.ptr(chan), opt(bmean), opt(bvar), _hip.stream_of(x))
        _hip.check(rc, "fpsg_dec1_fwd")
        ctx.save_for_backward(w1, x, pts, hlat, chan)
        ctx.cfg = (G, D, B, P, L, K, bool(training))
        if training:
            ctx.mark_non_differentiable(bmean, bvar)
            return out, bmean, bvar
        return out, None, None

    @staticmethod
    def backward(ctx, dout, _gm, _gv):
        if dout is None:
            return (None,) * 11
        from . import _hip
        lib = _hip.load()
        w1, x, pts, hlat, chan = ctx.saved_tensors
        G, D, B, P, L, K, training = ctx.cfg
        dev = x.device
        dout = dout.contiguous()
        T = lib.fpsg_dec1_tiles(D)
        dhlat = torch.empty((G, D, B), dtype=torch.float32, device=dev)
        gw = torch.empty((G, D, K), dtype=torch.float32, device=dev)       # latent columns by the GEMM below, point columns by K9
        dpts_part = torch.empty((G, T, 3, B * P), dtype=torch.float32, device=dev)
        dgamma = torch.empty((G * D,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((G * D,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.fpsg_dec1_bwd(_hip.ptr(dout), _hip.ptr(hlat), _hip.ptr(w1), K, L, _hip.ptr(pts), _hip.ptr(chan),
                                   G, D, B, P, 1 if training else 0, _hip.ptr(dhlat), _hip.ptr(gw), _hip.ptr(dpts_part),
                                   _hip.ptr(dgamma), _hip.ptr(dbeta), _hip.stream_of(x))
        _hip.check(rc, "fpsg_dec1_bwd")
        # d/dW[:, :L] = dhlat x  written straight into the stacked gradient (ldc = L + 3): no split / concat copies
        torch.bmm(dhlat, x.unsqueeze(0).expand(G, B, L), out=gw[..., :L])
        gb = dhlat.sum(dim=2, keepdim=True)
        gx = torch.matmul(dhlat.reshape(G * D, B).t(), w1.reshape(G * D, K)[:, :L])          # [B,L]
        gpts = dpts_part.sum(dim=1)
        return gw, gb, gx, gpts, dgamma, dbeta, None, None, None, None, None


class _DecoderLayer1Pair(torch.autograd.Function):
    """``_DecoderLayer1`` for several decodes side by side (training mode): ``x [B, L]`` holds the latents of all decodes
    in order, ``batches`` their sizes, ``pts`` / the result ``[G, ., B*P]`` their columns in the same order.  One latent
    GEMM for all of them, one K9 launch per decode (its BatchNorm statistics are that decode's), every later layer then
    runs once on the joint tensor.  Returns ``(out [G,D,B*P], stats [n, 2, G*D])``."""

    @staticmethod
    def forward(ctx, w1, b1, x, pts, gamma, beta, eps, P, batches):
        from . import _hip
        lib = _hip.load()
        G, D, K = w1.shape
        B, L = x.shape
        M = B * P
        ctx.set_materialize_grads(False)
        w1 = w1.contiguous()
        pts = pts.contiguous()
        hlat = torch.baddbmm(b1, w1[..., :L], x.t().unsqueeze(0).expand(G, L, B))          # [G,D,B]
        out = torch.empty((G, D, M), dtype=torch.float32, device=x.device)
        chan = torch.empty((len(batches), 4, G * D), dtype=torch.float32, device=x.device)
        stats = torch.empty((len(batches), 2, G * D), dtype=torch.float32, device=x.device)
        b0 = 0
        with torch.cuda.device(x.device):
            for i, Bi in enumerate(batches):
                rc = lib.fpsg_dec1_fwd_ld(_hip.ptr(hlat) + 4 * b0, B, _hip.ptr(w1), K, L, _hip.ptr(pts) + 4 * b0 * P, M,
                                          _hip.ptr(gamma), _hip.ptr(beta), None, None, G, D, Bi, P, 1, float(eps),
                                          _hip.ptr(out) + 4 * b0 * P, M, _hip.ptr(chan[i]), _hip.ptr(stats[i, 0]),
                                          _hip.ptr(stats[i, 1]), _hip.stream_of(x))
                _hip.check(rc, "fpsg_dec1_fwd_ld")
                b0 += Bi
        ctx.save_for_backward(w1, x, pts, hlat, chan)
        ctx.cfg = (G, D, B, P, L, K, tuple(batches))
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dout, _gstats):
        if dout is None:
            return (None,) * 9
        from . import _hip
        lib = _hip.load()
        w1, x, pts, hlat, chan = ctx.saved_tensors
        G, D, B, P, L, K, batches = ctx.cfg
        M = B * P
        dev = x.device
        dout = dout.contiguous()
        T = lib.fpsg_dec1_tiles(D)
        dhlat = torch.empty((G, D, B), dtype=torch.float32, device=dev)
        gw = torch.empty((G, D, K), dtype=torch.float32, device=dev)       # latent columns by the GEMM below, point columns by K9
        gpts = torch.empty((G, 3, M), dtype=torch.float32, device=dev)
        dgamma = torch.empty((G * D,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((G * D,), dtype=torch.float32, device=dev)
        b0 = 0
        with torch.cuda.device(dev):
            for i, Bi in enumerate(batches):
                part = torch.empty((G, T, 3, Bi * P), dtype=torch.float32, device=dev)
                rc = lib.fpsg_dec1_bwd_ld(_hip.ptr(dout) + 4 * b0 * P, M, _hip.ptr(hlat) + 4 * b0, B, _hip.ptr(w1), K, L,
                                          _hip.ptr(pts) + 4 * b0 * P, M, _hip.ptr(chan[i]), G, D, Bi, P, 1,
                                          1 if i else 0, _hip.ptr(dhlat) + 4 * b0, _hip.ptr(gw), _hip.ptr(part),
                                          _hip.ptr(dgamma), _hip.ptr(dbeta), _hip.stream_of(x))
                _hip.check(rc, "fpsg_dec1_bwd_ld")
                torch.sum(part, dim=1, out=gpts[:, :, b0 * P:(b0 + Bi) * P])
                b0 += Bi
        # d/dW[:, :L] = dhlat x  written straight into the stacked gradient (ldc = L + 3): no split / concat copies
        torch.bmm(dhlat, x.unsqueeze(0).expand(G, B, L), out=gw[..., :L])
        gb = dhlat.sum(dim=2, keepdim=True)
        gx = torch.matmul(dhlat.reshape(G * D, B).t(), w1.reshape(G * D, K)[:, :L])          # [B,L]
        return gw, gb, gx, gpts, dgamma, dbeta, None, None, None


class _StackFrozen(torch.autograd.Function):
    """``torch.stack(tensors)`` whose VALUES are taken from the step-scoped cache of ``winograd.weights_frozen`` when
    the same parameters were stacked earlier in the step (the parameters do not change between the episodes of an
    optimizer step: 247 MB of decoder weights are then copied once per step instead of once per episode); the
    gradient is unbound to the parameters as ``torch.stack``'s is."""

    @staticmethod
    def forward(ctx, key, *tensors):
        from . import winograd
        cache = winograd.frozen_cache()
        full_key = ("stack", key) + tuple(t.data_ptr() for t in tensors)
        if cache is not None and full_key in cache:
            return cache[full_key].detach()          # a new tensor on the cached storage: this node's output
        out = torch.stack([t.detach() for t in tensors])
        if cache is not None:
            cache[full_key] = out
            return out.detach()
        return out

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g.unbind(0))


class _RepeatFrozen(torch.autograd.Function):
    """``t.repeat_interleave(r, dim=0)`` whose VALUES come from the step-scoped cache of ``winograd.weights_frozen`` when
    the same tensor was repeated earlier in the step (a deformer's parameters serve its cluster's ``r`` patches: ten small
    copies per episode otherwise); the gradient is the sum over the ``r`` copies, as ``repeat_interleave``'s is."""

    @staticmethod
    def forward(ctx, t, r):
        from . import winograd
        ctx.r = r
        cache = winograd.frozen_cache()
        key = ("repeat", t.data_ptr(), tuple(t.shape), r)
        if cache is not None and key in cache:
            return cache[key].detach()
        out = t.detach().repeat_interleave(r, dim=0)
        if cache is not None:
            cache[key] = out
            return out.detach()
        return out

    @staticmethod
    def backward(ctx, g):
        return g.reshape(g.shape[0] // ctx.r, ctx.r, *g.shape[1:]).sum(1), None


class _RepeatAllFrozen(torch.autograd.Function):
    """``_RepeatFrozen`` for several ``[K, ...]`` tensors at once: the same cached values, but the backward sums the ``r``
    copies of ALL of them with one concatenation and one reduction (ten reductions of a few microseconds otherwise);
    the gradients returned are slices of that one result."""

    @staticmethod
    def forward(ctx, r, *ts):
        ctx.r = r
        ctx.shapes = [tuple(t.shape) for t in ts]
        return tuple(_RepeatFrozen.forward(ctx, t, r) for t in ts)

    @staticmethod
    def backward(ctx, *gs):
        r, shapes = ctx.r, ctx.shapes
        K = shapes[0][0]
        if any(g is None for g in gs):
            return (None,) + tuple(None if g is None else g.reshape(K, r, *sh[1:]).sum(1) for g, sh in zip(gs, shapes))
        total = torch.cat([g.reshape(K, r, -1) for g in gs], dim=2).sum(1)                 # [K, sum of the row sizes]
        out, o = [], 0
        for sh in shapes:
            n = 1
            for d in sh[1:]:
                n *= d
            out.append(total[:, o:o + n].reshape(sh) if len(sh) > 1 else total[:, o])
            o += n
        return (None,) + tuple(out)


def _repeat_all(ts, r):
    """``[t.repeat_interleave(r, dim=0) for t in ts]`` (all with the same leading size)."""
    ts = list(ts)
    if r == 1:
        return ts
    if all(t.is_cuda and t.requires_grad for t in ts) and torch.is_grad_enabled():
        return list(_RepeatAllFrozen.apply(r, *ts))
    return [_repeat_rows(t, r) for t in ts]


def _repeat_rows(t, r):
    if r == 1:
        return t
    if t.is_cuda and torch.is_grad_enabled() and t.requires_grad:
        return _RepeatFrozen.apply(t, r)
    return t.repeat_interleave(r, dim=0)


class _StackView(torch.autograd.Function):
    """``torch.stack(tensors)`` of tensors that ARE the consecutive rows of one contiguous block (a stack group of
    ``fpsg_amd.optim.layout_order`` inside the optimizer's flat parameter buffer): a view of that block, no copy; the
    gradient is unbound to the parameters as ``torch.stack``'s is."""

    @staticmethod
    def forward(ctx, *tensors):
        t0 = tensors[0].detach()
        return t0.as_strided((len(tensors),) + tuple(t0.shape), (t0.numel(),) + tuple(t0.stride()), t0.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return tuple(g.unbind(0))


_rows_of_one_block = {}      # (every tensor's address, shape) -> bool: the full check runs once per parameter layout


def _are_rows_of_one_block(tensors) -> bool:
    t0 = tensors[0]
    # the storage's identity and t0's place in it are part of the key: another model's parameters, allocated at the same
    # addresses after the first flat buffer was freed, must not inherit a cached "yes"
    key = (tuple(t.data_ptr() for t in tensors), t0.shape, t0.untyped_storage().data_ptr(), t0.storage_offset(),
           t0.untyped_storage().nbytes())
    hit = _rows_of_one_block.get(key)
    if hit is None:
        n, base, o0 = t0.numel(), t0.untyped_storage().data_ptr(), t0.storage_offset()
        hit = len(tensors) > 1 and all(
            t.shape == t0.shape and t.dtype == t0.dtype and t.is_contiguous()
            and t.untyped_storage().data_ptr() == base and t.storage_offset() == o0 + i * n
            for i, t in enumerate(tensors))
        if len(_rows_of_one_block) > 4096:
            _rows_of_one_block.clear()
        _rows_of_one_block[key] = hit
    return hit


class _BmmSplit(torch.autograd.Function):
    """``torch.bmm(w, h)`` for the patch MLPs' wide layers (``w [G, out, in]``, ``h [G, in, B*P]``: 1539 -> 769 and
    769 -> 384 of ``PrimitiveNode``, reference ``point_cloud_net.py:66-79``) on K10 (``FPSG_GEMM_SPLIT=1``, opt-in): forward
    ``w . h``, ``dh = w^T . g`` (the transposed weights are made once per optimizer step: ``winograd.weights_frozen``'s
    cache) and ``dw = g . h^T`` (the reduction over the B*P points split over workgroups)."""

    @staticmethod
    def forward(ctx, w, h):
        from .gemm_split import bmm_split
        w, h = w.contiguous(), h.contiguous()
        ctx.save_for_backward(w, h)
        return bmm_split(w, h, False)

    @staticmethod
    def backward(ctx, g):
        from . import winograd
        from .gemm_split import bmm_split
        w, h = ctx.saved_tensors
        g = g.contiguous()
        gw = gh = None
        if ctx.needs_input_grad[1]:
            cache = winograd.frozen_cache()
            key = ("bmm_wT", w.data_ptr(), tuple(w.shape))
            wT = cache.get(key) if cache is not None else None
            if wT is None:
                wT = w.transpose(1, 2).contiguous()
                if cache is not None:
                    cache[key] = wT
            gh = bmm_split(wT, g, False)
        if ctx.needs_input_grad[0]:
            gw = bmm_split(g, h, True)
        return gw, gh


class _BmmWideT(torch.autograd.Function):
    """``torch.bmm(w, h)`` for the patch MLPs' wide layers on the library's fp32 GEMMs, with the data gradient
    ``dh = w^T . g`` taken from a TRANSPOSED COPY of the weights whose rows start on 32-byte boundaries (``[G, in, out]``
    in a buffer of row stride ``out`` rounded up to 8 floats), made once per optimizer step
    (``winograd.weights_frozen``'s cache).  Autograd's own backward multiplies by ``w^T`` through its strides -- a
    transposed-operand kernel reading 1539-float rows (odd: 4-byte aligned) -- 1461 us per episode at 1539 -> 769; the
    plain product on the aligned copy takes 1317 us (``profiles/r05/decoder_weight_rows.txt``).  Same arithmetic (fp32
    library GEMM), another kernel's summation order.  ``FPSG_DECODER_WT=0``: autograd's form (A/B)."""

    @staticmethod
    def _aligned(w, transposed):
        """``w`` (``transposed``: ``w^T``) in a buffer whose rows start on 32-byte boundaries, once per optimizer step."""
        from . import winograd
        cache = winograd.frozen_cache()
        key = ("bmm_wT_rows" if transposed else "bmm_w_rows", w.data_ptr(), tuple(w.shape), tuple(w.stride()))
        a = cache.get(key) if cache is not None else None
        if a is None:
            src = w.transpose(1, 2) if transposed else w
            G, rows, cols = src.shape
            a = torch.empty((G, rows, (cols + 7) // 8 * 8), dtype=w.dtype, device=w.device)[:, :, :cols]
            a.copy_(src)
            if cache is not None:
                cache[key] = a
        return a

    @staticmethod
    def forward(ctx, w, h):
        ctx.save_for_backward(w, h)
        return torch.bmm(w, h)      # (the forward from an aligned copy too: measured, no difference -- profiles/r05/decoder_weight_rows.txt)

    @staticmethod
    def backward(ctx, g):
        w, h = ctx.saved_tensors
        gw = gh = None
        if ctx.needs_input_grad[1]:
            wT = _BmmWideT._aligned(w, True)
            gh = torch.bmm(wT, g)
        if ctx.needs_input_grad[0]:
            gw = torch.bmm(g, h.transpose(1, 2))
        return gw, gh


_WT_MIN_COLUMNS = 2048      # below 16 clouds x 128 points the copy costs more than the aligned product saves (one-shot episodes)


def _bmm_wide(w, h):
    """``torch.bmm(w, h)``; layers of at least 128 inputs and outputs through K10 when ``FPSG_GEMM_SPLIT=1``, otherwise
    (training, on the GPU) with the data gradient from an aligned transposed copy of the weights (``_BmmWideT``)."""
    from . import gemm_split
    if gemm_split.enabled() and w.is_cuda and min(w.shape[1], w.shape[2]) >= 128:
        return _BmmSplit.apply(w, h)
    if (w.is_cuda and torch.is_grad_enabled() and h.requires_grad and min(w.shape[1], w.shape[2]) >= 128
            and h.shape[2] >= _WT_MIN_COLUMNS and os.environ.get("FPSG_DECODER_WT", "1") != "0"):
        return _BmmWideT.apply(w, h)
    return torch.bmm(w, h)


def _stack(key, tensors):
    tensors = list(tensors)
    if tensors[0].is_cuda and _are_rows_of_one_block(tensors):
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            return _StackView.apply(*tensors)
        return _StackView.forward(None, *tensors)
    if tensors[0].is_cuda and torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return _StackFrozen.apply(key, *tensors)
    return torch.stack(tensors)


class _LazySplit:
    """``(w[..., :L], w[..., L:], b)`` computed once, when first asked for."""

    def __init__(self, w, b, L):
        self._args, self._val = (w, b, L), None

    def __call__(self):
        if self._val is None:
            w, b, L = self._args
            self._val = _SplitLast.apply(w, L) + (b,)
        return self._val


def _stack_affine(bns, calls_per_bn):
    """gamma / beta of ``bns`` stacked to ``[G*C]`` for ``_group_batch_norm``."""
    gamma, beta = _stack("bn.weight", [b.weight for b in bns]), _stack("bn.bias", [b.bias for b in bns])
    gamma, beta = _repeat_rows(gamma, calls_per_bn), _repeat_rows(beta, calls_per_bn)
    return gamma.reshape(-1), beta.reshape(-1)


def _group_batch_norm(h, bns, calls_per_bn, act, affine=None, pre_bias=None):
    """BatchNorm over the last axis of ``h [G, C, M]`` with an independent set of statistics
    per group ``g`` (one reference ``BatchNorm1d`` call each), then ``act``.

    ``pre_bias`` ``[G*C]``: the bias of the layer that produced ``h``, added inside the BatchNorm pass (K5) instead
    of by the GEMM (``baddbmm`` with a broadcast bias first copies it over the whole output, then reads it back).
    ``bns``: the ``len(bns) * calls_per_bn == G`` modules owning the affine parameters and
    running statistics; module ``i`` serves groups ``i*calls_per_bn .. (i+1)*calls_per_bn-1``,
    i.e. it is "called" ``calls_per_bn`` times in that order, as in the reference loop, and
    its running statistics receive that many sequential momentum updates."""
    G, C, M = h.shape
    r = calls_per_bn
    gamma, beta = affine if affine is not None else _stack_affine(bns, r)
    x = h.reshape(1, G * C, M)
    fuse = "relu" if act is F.relu else None       # BatchNorm + ReLU as one pass (K5) on the GPU
    post = (lambda t: t) if act is F.relu else act
    if bns[0].training:
        y, mean, var = batch_norm_act(x, gamma, beta, None, None, True, 1.0, bns[0].eps, fuse, return_stats=True,
                                      pre_bias=pre_bias)
        with torch.no_grad():
            _update_running(bns, r, mean, var)
    else:
        rm = torch.stack([b.running_mean for b in bns]).repeat_interleave(r, dim=0).reshape(G * C)
        rv = torch.stack([b.running_var for b in bns]).repeat_interleave(r, dim=0).reshape(G * C)
        y = batch_norm_act(x, gamma, beta, rm, rv, False, 0.0, bns[0].eps, fuse, pre_bias=pre_bias)
    return post(y.reshape(G, C, M))


def _group_batch_norm_rows(h, bns, calls_per_bn, act, affine, pre_bias, seg_lens):
    """``_group_batch_norm`` (training mode) for several forward passes side by side: the column segments
    ``seg_lens`` of ``h [G, C, M]`` are consecutive passes, each normalised with its own batch statistics (K5, one
    launch) and followed by its own running-statistics updates, in order."""
    G, C, M = h.shape
    gamma, beta = affine
    fuse = "relu" if act is F.relu else None
    post = (lambda t: t) if act is F.relu else act
    y, stats = batch_norm_act_rows(h, gamma, beta, seg_lens, bns[0].eps, fuse, pre_bias=pre_bias)
    with torch.no_grad():
        _update_running_calls(bns, calls_per_bn, stats)
    return post(y)


class PCDecoder(nn.Module):
    """AtlasNet-style decoder: ``num_clusters`` clusters x ``num_nodes`` patches;
    ``forward(hidden[B, bottleneck]) -> [B, num_pts, 3]`` (contiguous).

    ``batched=True`` (default) evaluates all ``clusters x nodes`` patch MLPs together: the
    per-patch weights are stacked at run time and every layer is ONE batched GEMM + ONE
    grouped BatchNorm instead of 16 small ones (the looped form is launch-bound on MI355X:
    ~1000 kernels of a few microseconds per decode).  Same arithmetic per patch, same
    per-call BatchNorm statistics and running-statistics updates, same state dict."""

    def __init__(self, conf, num_pts: int = 2048, batched: bool = True):
        super().__init__()
        self.conf = conf
        self.device = conf.device
        self.num_nodes = conf.num_nodes
        self.num_clusters = conf.num_clusters
        self.num_pts_per_cluster = num_pts // self.num_clusters
        self.batched = batched
        self.cluster_pool = nn.ModuleList([
            PrimitiveCluster(conf, MLPDeformer(conf), self.num_pts_per_cluster, self.num_nodes)
            for _ in range(self.num_clusters)
        ])
        self._tag_stack_groups()

    def _tag_stack_groups(self):
        """Marks the parameters ``pack_parameters`` stacks -- the same tensor of every deformer, of every node -- as
        stack groups (``fpsg_amd.optim.layout_order``): the flat optimizer then stores each group as one contiguous
        ``[n, ...]`` block, and stacking it is a view.  The modules, their parameters and the state-dict keys stay
        the reference's (``cluster_pool.<i>.deformer.*``, ``cluster_pool.<i>.node_pool.<j>.*``)."""
        clusters = list(self.cluster_pool)
        families = {"deformer": [c.deformer for c in clusters],
                    "node": [n for c in clusters for n in c.node_pool]}
        for fam, mods in families.items():
            names = [name for name, _ in mods[0].named_parameters()]
            for name in names:
                members = [dict(m.named_parameters())[name] for m in mods]
                for i, p in enumerate(members):
                    p._fpsg_stack = ((id(self), fam, name), i, len(members))

    def sample_grids(self, batch: int, device, generator=None):
        """Nested list ``[cluster][node] -> [B, ori_dim, P]`` drawn in the reference's order."""
        return [c.sample_grids(batch, device, generator) for c in self.cluster_pool]

    def forward(self, hidden_feat, grid=None, generator=None, pack=None):
        """``grid`` (optional): nested list ``[cluster][node] -> [B, ori_dim, P]``.
        ``pack`` (optional, batched form): the result of ``pack_parameters()``, shared by the
        decodes of one episode."""
        if self.batched:
            return self._forward_batched(hidden_feat, grid, generator, pack)
        return self._forward_looped(hidden_feat, grid, generator)

    def pair_ready(self, hidden_a, hidden_b) -> bool:
        """``forward_pair`` has its joint form for these inputs: the batched decoder in training mode on the GPU, ReLU
        patch MLPs, K9's patch sizes.  ``FPSG_DECODE_PAIR=0`` keeps the two separate passes (A/B measurements)."""
        import os
        from . import fused_bn
        c0 = self.cluster_pool[0]
        P = c0.pts_per_node
        Ba, Bb = hidden_a.shape[0], hidden_b.shape[0]
        # K5's row form (batch_norm_act_rows) has no library fallback: it needs the fused BatchNorm switched on
        # (FPSG_FUSED_BN), a joint row of at least 64 columns in multiples of 4, segment starts on multiples of 4 and
        # segments within the kernel's row limit -- otherwise the two decodes run as two passes
        rows_ok = (fused_bn.fused_enabled() and (Ba + Bb) * P >= fused_bn._MIN_ROW and (Ba * P) % 4 == 0
                   and (Bb * P) % 4 == 0 and max(Ba, Bb) * P <= fused_bn.ROWS_SEGMENT_MAX)
        return (rows_ok and self.batched and self.training and hidden_a.is_cuda and hidden_a.dtype == torch.float32
                and hidden_a.dim() == 2 and hidden_b.dim() == 2 and hidden_a.shape[1] == hidden_b.shape[1]
                and os.environ.get("FPSG_DECODE_PAIR", "1") != "0" and self.conf.raw_dim == 3
                and c0.deformer.activation is F.relu and c0.node_pool[0].activation is F.relu
                and _layer1_fused_ok(hidden_a, hidden_a, max(hidden_a.shape[0], hidden_b.shape[0]), P, F.relu))

    def forward_pair(self, hidden_a, hidden_b, generator=None, pack=None, grids=None):
        """``forward(hidden_a)`` then ``forward(hidden_b)`` -- the two decodes of an episode with intra-reconstruction
        (queries, then supports) -- returned as ONE tensor ``[Ba + Bb, num_pts, 3]`` (a's clouds first).  Where
        ``pair_ready``: every GEMM runs once on the two decodes' columns side by side, while each BatchNorm call keeps
        its own batch statistics and its own running-statistics update, in the order of two separate passes (K5 over
        column segments, K9 per decode); the shared parameters receive one gradient instead of two to be added.
        ``grids`` (optional): ``(grid_a, grid_b)``, each as ``forward``'s ``grid``."""
        if not self.pair_ready(hidden_a, hidden_b):
            ga, gb = grids if grids is not None else (None, None)
            return torch.cat([self.forward(hidden_a, ga, generator, pack), self.forward(hidden_b, gb, generator, pack)])
        if pack is None:
            pack = self.pack_parameters()
        clusters = list(self.cluster_pool)
        K, R = len(clusters), self.num_nodes
        G = K * R
        P = clusters[0].pts_per_node
        act = clusters[0].deformer.activation
        batches = (hidden_a.shape[0], hidden_b.shape[0])
        B = sum(batches)
        seg_lens = tuple(b * P for b in batches)
        x = torch.cat([hidden_a, hidden_b])
        L = x.shape[1]
        dim = clusters[0].template[0].dim
        square = type(clusters[0].template[0]).__name__ == "SquareTemplate"
        if grids is None or (grids[0] is None and grids[1] is None):
            # both decodes' patch samples from one draw (i.i.d. either way): one generator launch, one permuting copy
            g = torch.empty((G, B, dim, P), dtype=torch.float32, device=x.device)
            g = g.uniform_(0, 1, generator=generator) if square else g.normal_(0, 1, generator=generator)
            h = g.permute(0, 2, 1, 3).reshape(G, dim, B * P)                            # [G,dim,B*P]
        else:
            cols = []
            for i, b in enumerate(batches):
                if grids[i] is not None:
                    g = torch.stack([t for per_cluster in grids[i] for t in per_cluster])  # [G,b,dim,P]
                else:
                    g = torch.empty((G, b, dim, P), dtype=torch.float32, device=x.device)
                    g = g.uniform_(0, 1, generator=generator) if square else g.normal_(0, 1, generator=generator)
                cols.append(g.permute(0, 2, 1, 3).reshape(G, g.size(2), b * P))
            h = torch.cat(cols, dim=2)                                                  # [G,dim,B*P]

        defs = [c.deformer for c in clusters]
        w, b = pack["d1"]
        h = _group_batch_norm_rows(torch.bmm(w, h), [d.bn1 for d in defs], R, act, pack["dbn1"], b.reshape(-1), seg_lens)
        w, b = pack["d2"]
        h = _group_batch_norm_rows(torch.bmm(w, h), [d.bn2 for d in defs], R, act, pack["dbn2"], b.reshape(-1), seg_lens)
        w, b = pack["d3"]
        pts = torch.tanh(torch.baddbmm(b, w, h))                                        # [G,raw,B*P]

        nodes = [n for c in clusters for n in c.node_pool]
        w1, b1 = pack["n1"]
        if w1.size(2) - self.conf.raw_dim != L:
            raise ValueError(f"PCDecoder: hidden size {L} does not match the first layer ({w1.size(2) - self.conf.raw_dim} latent columns)")
        bn1s = [n.bn1 for n in nodes]
        gamma, beta = pack["nbn1"]
        h, stats = _DecoderLayer1Pair.apply(w1, b1, x, pts, gamma, beta, bn1s[0].eps, P, batches)
        with torch.no_grad():
            _update_running_calls(bn1s, 1, stats)
        w, b = pack["n2"]
        h = _group_batch_norm_rows(_bmm_wide(w, h), [n.bn2 for n in nodes], 1, act, pack["nbn2"], b.reshape(-1), seg_lens)
        w, b = pack["n3"]
        h = _group_batch_norm_rows(_bmm_wide(w, h), [n.bn3 for n in nodes], 1, act, pack["nbn3"], b.reshape(-1), seg_lens)
        w, b = pack["n4"]
        out = torch.tanh(torch.baddbmm(b, w, h))                                        # [G,3,B*P]
        return out.view(G, 3, B, P).permute(2, 0, 3, 1).reshape(B, G * P, 3).contiguous()

    def pack_parameters(self):
        """The per-patch weights / biases / BatchNorm affine parameters stacked for the batched
        form.  An episode with intra-reconstruction decodes twice (queries, supports): stacking
        once and passing the pack to both calls halves the stacking copies and, in the backward,
        sums the two decodes' gradients on the 24 stacked tensors instead of leaving autograd a
        second, accumulating gradient for each of the 264 per-patch parameters."""
        if not self.batched:
            return None
        # Without autograd (the evaluation loop) inside a ``winograd.weights_frozen`` block the parameters do not change:
        # the pack is made by the block's first item and reused (35 stacking launches per evaluated item otherwise; a
        # hipGraph captured inside a ``constant`` block reads the entry and holds none of them).
        cache = key = None
        if not torch.is_grad_enabled() and os.environ.get("FPSG_EVAL_PACK_CACHE", "1") != "0":
            from . import winograd
            cache = winograd.frozen_cache()
            first = next(self.parameters())
            key = ("decoder_pack", id(self), first.data_ptr(), first.device)
            if cache is not None and key in cache:
                return cache[key]
        pack = self._build_pack()
        if cache is not None and not (first.is_cuda and torch.cuda.is_current_stream_capturing()):
            if callable(pack.get("n1_split")):
                pack["n1_split"]()        # no lazily made tensor may first appear inside a later capture
            cache[key] = pack
        return pack

    def _build_pack(self):
        clusters = list(self.cluster_pool)
        R = self.num_nodes
        defs = [c.deformer for c in clusters]
        nodes = [n for c in clusters for n in c.node_pool]

        def stack_w(mods, name):
            w = _stack(name + ".w", [getattr(m, name).weight.squeeze(-1) for m in mods])   # [n,out,in]
            b = _stack(name + ".b", [getattr(m, name).bias for m in mods])                 # [n,out]
            return w, b

        # the deformers' ten tensors serve their cluster's R patches: repeated together (one backward reduction)
        dw = [t for i in (1, 2, 3) for t in stack_w(defs, f"conv{i}")]
        daff = [_stack(f"bn{i}.{a}", [getattr(getattr(d, f"bn{i}"), a) for d in defs]) for i in (1, 2) for a in ("weight", "bias")]
        rep = _repeat_all(dw + daff, R)
        pack = {f"d{i}": (rep[2 * (i - 1)], rep[2 * (i - 1) + 1].unsqueeze(-1)) for i in (1, 2, 3)}
        pack.update({f"n{i}": (lambda wb: (wb[0], wb[1].unsqueeze(-1)))(stack_w(nodes, f"conv{i}")) for i in (1, 2, 3, 4)})
        # first node layer: latent and point columns of the stacked weight, split once per pack
        w1, b1 = pack["n1"]
        L = w1.size(2) - self.conf.raw_dim
        pack["n1_split"] = _LazySplit(w1, b1, L)      # latent / point columns, split on first use (library path only)
        pack["dbn1"] = (rep[6].reshape(-1), rep[7].reshape(-1))
        pack["dbn2"] = (rep[8].reshape(-1), rep[9].reshape(-1))
        for i in (1, 2, 3):
            pack[f"nbn{i}"] = _stack_affine([getattr(n, f"bn{i}") for n in nodes], 1)
        return pack

    def _forward_looped(self, hidden_feat, grid=None, generator=None):
        outs = []
        for ci, cluster in enumerate(self.cluster_pool):
            outs.append(cluster(hidden_feat, None if grid is None else grid[ci], generator))
        return torch.cat(outs, dim=2).transpose(1, 2).contiguous()

    def _forward_batched(self, x, grid=None, generator=None, pack=None):
        if pack is None:
            pack = self.pack_parameters()
        clusters = list(self.cluster_pool)
        K, R = len(clusters), self.num_nodes
        G = K * R
        B, L = x.shape
        P = clusters[0].pts_per_node
        act = clusters[0].deformer.activation
        if grid is None:
            dim = clusters[0].template[0].dim
            if type(clusters[0].template[0]).__name__ == "SquareTemplate":
                g = torch.empty((G, B, dim, P), dtype=torch.float32, device=x.device).uniform_(0, 1, generator=generator)
            else:
                g = torch.empty((G, B, dim, P), dtype=torch.float32, device=x.device).normal_(0, 1, generator=generator)
        else:
            g = torch.stack([t for per_cluster in grid for t in per_cluster])          # [G,B,dim,P]
        h = g.permute(0, 2, 1, 3).reshape(G, g.size(2), B * P)                          # [G,dim,B*P]

        # ---- per-cluster deformers, applied to each of the cluster's R patches
        defs = [c.deformer for c in clusters]
        w, b = pack["d1"]
        h = _group_batch_norm(torch.bmm(w, h), [d.bn1 for d in defs], R, act, pack["dbn1"], b.reshape(-1))
        w, b = pack["d2"]
        h = _group_batch_norm(torch.bmm(w, h), [d.bn2 for d in defs], R, act, pack["dbn2"], b.reshape(-1))
        w, b = pack["d3"]
        pts = torch.tanh(torch.baddbmm(b, w, h))                                        # [G,raw,B*P]

        # ---- the G patch MLPs; first layer split into latent and point parts
        nodes = [n for c in clusters for n in c.node_pool]
        w1, b1 = pack["n1"]                                                             # [G,D,L+raw], [G,D,1]
        D = w1.size(1)
        if w1.size(2) - self.conf.raw_dim != L:
            raise ValueError(f"PCDecoder: hidden size {L} does not match the first layer ({w1.size(2) - self.conf.raw_dim} latent columns)")
        bn1s = [n.bn1 for n in nodes]
        if self.conf.raw_dim == 3 and _layer1_fused_ok(x, pts, B, P, act):
            # K9: latent GEMM + one pass; the [G,D,B*P] pre-BatchNorm tensor is never stored
            gamma, beta = pack["nbn1"]
            training = bn1s[0].training
            rm = rv = None
            if not training:
                rm = torch.stack([b_.running_mean for b_ in bn1s]).reshape(-1)
                rv = torch.stack([b_.running_var for b_ in bn1s]).reshape(-1)
            h, bmean, bvar = _DecoderLayer1.apply(w1, b1, x, pts, gamma, beta, rm, rv, training, bn1s[0].eps, P)
            if training:
                with torch.no_grad():
                    _update_running(bn1s, 1, bmean, bvar)
        else:
            w_lat, w_pts, b = pack["n1_split"]() if callable(pack["n1_split"]) else pack["n1_split"]   # [G,D,L], [G,D,raw]
            h_lat = torch.matmul(w_lat, x.t()) + b                                      # [G,D,B]
            h = torch.bmm(w_pts, pts).view(G, D, B, P) + h_lat.unsqueeze(-1)
            h = _group_batch_norm(h.view(G, D, B * P), bn1s, 1, act, pack["nbn1"])
        w, b = pack["n2"]
        h = _group_batch_norm(_bmm_wide(w, h), [n.bn2 for n in nodes], 1, act, pack["nbn2"], b.reshape(-1))
        w, b = pack["n3"]
        h = _group_batch_norm(_bmm_wide(w, h), [n.bn3 for n in nodes], 1, act, pack["nbn3"], b.reshape(-1))
        w, b = pack["n4"]
        out = torch.tanh(torch.baddbmm(b, w, h))                                        # [G,3,B*P]
        return out.view(G, 3, B, P).permute(2, 0, 3, 1).reshape(B, G * P, 3).contiguous()
