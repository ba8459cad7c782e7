"""DGCNN point-cloud encoder on HIP kernels: kNN graph + EdgeConv.

Mirrors reference ``src/dgcnn/model.py``: ``knn :13-20``, ``get_graph_feature :23-42``,
``DGCNNfeat :45-88`` (same constructor, same ``conv1..conv5`` ``nn.Sequential`` layout, hence
the same state-dict keys ``conv<i>.0.weight``, ``conv<i>.1.{weight,bias,running_*}``).

  * ``knn`` -> ``fpsg_knn`` (K3): MFMA distance tiles + in-LDS top-k; the [B,N,N] matrix of
    the reference is never materialised.  Returns int64 ``[B,N,k]`` like ``topk``.
  * ``get_graph_feature`` -> ``fpsg_edge_feature_fwd/bwd`` (K4a): one gather pass writing
    ``[B,2C,N,k]`` (drop-in form; no hard-coded ``torch.device('cuda')``: the device is the
    input's).

There is no CPU path for either (raises on CPU tensors).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _hip


def knn_int32(x: torch.Tensor, k: int) -> torch.Tensor:
    """``x [B,C,N]`` fp32 on a ROCm device -> int32 ``[B,N,k]`` neighbour indices."""
    if x.dim() != 3:
        raise ValueError(f"expected x [B,C,N], got {tuple(x.shape)}")
    x = _hip.dev_tensor(x.detach() if x.requires_grad else x, torch.float32, "x")
    B, C, N = x.shape
    if not 0 < k <= min(N, 64):
        raise ValueError(f"k={k} must be in [1, min(N, 64)] (N={N})")
    idx = torch.empty((B, N, k), dtype=torch.int32, device=x.device)
    ws = torch.empty((B, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _hip.load().fpsg_knn(_hip.ptr(x), B, C, N, k, _hip.ptr(idx), _hip.ptr(ws),
                                  _hip.stream_of(x))
    _hip.check(rc, "fpsg_knn")
    return idx


def knn(x: torch.Tensor, k: int) -> torch.Tensor:
    """Reference signature: ``[B,C,N] -> LongTensor [B,N,k]`` (self first, nearest first)."""
    return knn_int32(x.contiguous(), k).long()


class _EdgeFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        _hip.dev_tensor(x, torch.float32, "x")
        _hip.dev_tensor(idx, torch.int32, "idx")
        B, C, N = x.shape
        k = idx.size(2)
        if idx.shape != (B, N, k):
            raise ValueError(f"idx {tuple(idx.shape)} does not match x {tuple(x.shape)}")
        out = torch.empty((B, 2 * C, N, k), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _hip.load().fpsg_edge_feature_fwd(_hip.ptr(x), _hip.ptr(idx), B, C, N, k,
                                                   _hip.ptr(out), _hip.stream_of(x))
        _hip.check(rc, "fpsg_edge_feature_fwd")
        ctx.save_for_backward(idx)
        ctx.dims = (B, C, N, k)
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        B, C, N, k = ctx.dims
        gout = gout.contiguous()
        gx = torch.empty((B, C, N), dtype=torch.float32, device=gout.device)
        with torch.cuda.device(gout.device):
            rc = _hip.load().fpsg_edge_feature_bwd(_hip.ptr(gout), _hip.ptr(idx), B, C, N, k,
                                                   _hip.ptr(gx), _hip.stream_of(gout))
        _hip.check(rc, "fpsg_edge_feature_bwd")
        return gx, None


def get_graph_feature(x: torch.Tensor, k: int = 20, idx: torch.Tensor | None = None) -> torch.Tensor:
    """Reference signature: ``x [B,C,N]`` (any trailing dims are flattened into N as the
    reference's ``view`` does) -> ``[B,2C,N,k]`` = ``cat(x_j - x_i, x_i)``."""
    B, C = x.size(0), x.size(1)
    x = x.reshape(B, C, -1).contiguous()
    if idx is None:
        idx32 = knn_int32(x, k)
    else:
        idx32 = idx.to(torch.int32).contiguous()
    return _EdgeFeature.apply(x, idx32)


def _edge_block(c_in: int, c_out: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=1, bias=False), nn.BatchNorm2d(c_out),
                         nn.LeakyReLU(negative_slope=0.2))


class DGCNNfeat(nn.Module):
    """4 EdgeConv layers (k=20) -> concat -> 1x1 conv -> max||avg pool: ``[B,3,N] -> [B,1024]``."""

    def __init__(self, embeding_dim: int = 1024, num_neighbors: int = 20, dual_pool: bool = True):
        super().__init__()
        self.dual_flag = dual_pool
        self.emb_dims = embeding_dim // 2 if dual_pool else embeding_dim
        self.k = num_neighbors
        self.conv1 = _edge_block(6, 64)
        self.conv2 = _edge_block(128, 64)
        self.conv3 = _edge_block(128, 128)
        self.conv4 = _edge_block(256, 256)
        self.conv5 = nn.Sequential(nn.Conv1d(512, self.emb_dims, kernel_size=1, bias=False),
                                   nn.BatchNorm1d(self.emb_dims), nn.LeakyReLU(negative_slope=0.2))

    def _edgeconv(self, block: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
        return block(get_graph_feature(x, k=self.k)).amax(dim=-1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x1 = self._edgeconv(self.conv1, x)
        x2 = self._edgeconv(self.conv2, x1)
        x3 = self._edgeconv(self.conv3, x2)
        x4 = self._edgeconv(self.conv4, x3)
        h = self.conv5(torch.cat((x1, x2, x3, x4), dim=1))
        if self.dual_flag:
            return torch.cat((h.amax(dim=2), h.mean(dim=2)), dim=1)
        return h.amax(dim=2)
