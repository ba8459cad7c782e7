#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by IMPORTING the reference's own Python
modules from /root/reference/src (possible only in the build container; the GPU box never
sees the reference).  Only data is written: seeded inputs and the reference's outputs.

    python tests/golden/make_golden.py

What runs from the reference:
  * pointnet.model.PointNetfeat / STN3d with the shipped checkpoint
    checkpoint/pretrain_pointnet/pretrained_pcencoder_pointnet.pt (copied here as a data
    fixture; MIT-licensed repository);
  * dgcnn.model.knn, get_graph_feature, DGCNNfeat.  get_graph_feature hard-codes
    torch.device('cuda') (dgcnn/model.py:29); the reference function is executed unchanged
    with that one call redirected to the CPU device through a proxy of the `torch` name in
    its module namespace;
  * datasets.utils.extract_episode / EpisodicBatchSampler / SequentialBatchSampler random
    streams under torch.manual_seed.
Not importable here (pymesh / neuralnet_pytorch / kaolin / torchvision absent): models.*
-- the decoder and the episode loss are pinned structurally (state-dict keys, parameter
counts, algebraic identities) instead.
"""
import os
import shutil
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
OUT = os.path.dirname(os.path.abspath(__file__))

from pointnet.model import PointNetfeat, STN3d  # noqa: E402
import dgcnn.model as dgm  # noqa: E402
from datasets.utils import EpisodicBatchSampler, SequentialBatchSampler, extract_episode  # noqa: E402


def pointnet_goldens():
    ckpt = os.path.join(REF, "checkpoint/pretrain_pointnet/pretrained_pcencoder_pointnet.pt")
    shutil.copyfile(ckpt, os.path.join(OUT, "pretrained_pcencoder_pointnet.pt"))
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    pref = "pc_encoder.pointnet_feat_extractor."
    inner = {k[len(pref):]: v for k, v in sd.items()}
    out = {}
    for tag, shape in (("a", (4, 3, 256)), ("b", (2, 3, 2048))):
        g = torch.Generator().manual_seed(100 + len(tag) + shape[2])
        x = torch.randn(*shape, generator=g) * 0.5
        out[f"x_{tag}"] = x.numpy()
        for mode in ("eval", "train"):
            m = PointNetfeat()
            m.load_state_dict(inner)
            m.train(mode == "train")
            with torch.no_grad():
                feat, trans, tf = m(x)
            assert tf is None
            out[f"feat_{tag}_{mode}"] = feat.numpy()
            out[f"trans_{tag}_{mode}"] = trans.numpy()
            if mode == "train":  # running statistics after one training-mode forward
                out[f"bn3_mean_{tag}_train"] = m.bn3.running_mean.numpy().copy()
                out[f"stn_bn5_var_{tag}_train"] = m.stn.bn5.running_var.numpy().copy()
            s = STN3d()
            s.load_state_dict({k[4:]: v for k, v in inner.items() if k.startswith("stn.")})
            s.train(mode == "train")
            with torch.no_grad():
                out[f"stn_{tag}_{mode}"] = s(x).numpy()
    np.savez_compressed(os.path.join(OUT, "pointnet_goldens.npz"), **out)
    print("pointnet_goldens.npz", {k: v.shape for k, v in out.items()})


class _TorchOnCPU:
    """`torch` as seen by dgcnn/model.py, with torch.device(...) pinned to the CPU."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def device(*_a, **_k):
        return torch.device("cpu")


def dgcnn_goldens():
    out = {}
    for tag, (B, C, N) in (("c3_n256", (2, 3, 256)), ("c3_n2048", (2, 3, 2048)),
                           ("c64_n256", (2, 64, 256)), ("c64_n2048", (1, 64, 2048))):
        g = torch.Generator().manual_seed(7 + C + N)
        x = torch.randn(B, C, N, generator=g)
        if C == 3:
            x = x / x.norm(dim=1, keepdim=True).amax(dim=2, keepdim=True)
        out[f"knn_x_{tag}"] = x.numpy()
        out[f"knn_idx_{tag}"] = dgm.knn(x, 20).numpy().astype(np.int32)
    dgm.torch = _TorchOnCPU()
    try:
        g = torch.Generator().manual_seed(21)
        x = torch.randn(2, 5, 96, generator=g)
        out["ggf_x"] = x.numpy()
        out["ggf_out"] = dgm.get_graph_feature(x, k=20).numpy()
        torch.manual_seed(33)
        net = dgm.DGCNNfeat()
        # non-trivial BN affine parameters (both signs) so that the fused max/min selection
        # of the HIP path is exercised
        with torch.no_grad():
            for mod in net.modules():
                if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                    mod.weight.copy_(torch.randn(mod.weight.shape) * 0.7)
                    mod.bias.copy_(torch.randn(mod.bias.shape) * 0.1)
        torch.save(net.state_dict(), os.path.join(OUT, "dgcnn_state.pt"))
        g = torch.Generator().manual_seed(22)
        pts = torch.randn(3, 3, 160, generator=g)
        pts = pts / pts.norm(dim=1, keepdim=True).amax(dim=2, keepdim=True)
        out["dgcnn_x"] = pts.numpy()
        for mode in ("eval", "train"):
            net2 = dgm.DGCNNfeat()
            net2.load_state_dict(torch.load(os.path.join(OUT, "dgcnn_state.pt"), weights_only=True))
            net2.train(mode == "train")
            with torch.no_grad():
                out[f"dgcnn_feat_{mode}"] = net2(pts).numpy()
            if mode == "train":
                out["dgcnn_conv4_bn_mean_train"] = net2.conv4[1].running_mean.numpy().copy()
                out["dgcnn_conv4_bn_var_train"] = net2.conv4[1].running_var.numpy().copy()
    finally:
        dgm.torch = torch
    np.savez_compressed(os.path.join(OUT, "dgcnn_goldens.npz"), **out)
    print("dgcnn_goldens.npz", {k: v.shape for k, v in out.items()})


def episode_goldens():
    out = {}
    torch.manual_seed(0)
    out["episodic_10_1_6"] = torch.stack(list(EpisodicBatchSampler(10, 1, 6))).numpy()
    out["episodic_40_3_4"] = torch.stack(list(EpisodicBatchSampler(40, 3, 4))).numpy()
    out["sequential_5"] = torch.stack(list(SequentialBatchSampler(5))).numpy()
    torch.manual_seed(1)
    d = {"class": "chair", "img_data": torch.arange(12.0).view(12, 1), "pc_data": torch.arange(12.0).view(12, 1) + 100}
    eps = [extract_episode(4, 1, d) for _ in range(5)]
    out["extract_xs"] = torch.stack([e["xs"].view(-1) for e in eps]).numpy()
    out["extract_xq"] = torch.stack([e["xq"].view(-1) for e in eps]).numpy()
    out["extract_pcs"] = torch.stack([e["pcs"].view(-1) for e in eps]).numpy()
    out["extract_tmp"] = np.array([e["tmp"] for e in eps])
    np.savez_compressed(os.path.join(OUT, "episode_streams.npz"), **out)
    print("episode_streams.npz", {k: v.shape for k, v in out.items()})


def _rows(t):
    """Large gradients are stored as their first 64 rows (fixtures stay small)."""
    return t[:64] if t.numel() > 65536 else t


def gradient_goldens():
    """Backward passes of the reference modules (training mode, CPU autograd): gradients of
    ``sum(feat * w)`` with respect to the input and to a spread of parameters -- what the
    HIP path's fused BatchNorm+max / EdgeConv backward kernels must reproduce."""
    out = {}
    ckpt = os.path.join(REF, "checkpoint/pretrain_pointnet/pretrained_pcencoder_pointnet.pt")
    sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    pref = "pc_encoder.pointnet_feat_extractor."
    inner = {k[len(pref):]: v for k, v in sd.items()}
    for tag, shape in (("a", (4, 3, 256)), ("b", (2, 3, 2048))):
        g = torch.Generator().manual_seed(100 + len(tag) + shape[2])
        x = (torch.randn(*shape, generator=g) * 0.5).requires_grad_()     # same inputs as pointnet_goldens
        w = torch.randn(shape[0], 1024, generator=g)
        out[f"pn_w_{tag}"] = w.numpy()
        m = PointNetfeat()
        m.load_state_dict(inner)
        m.train()
        feat, _, _ = m(x)
        (feat * w).sum().backward()
        out[f"pn_gx_{tag}"] = x.grad.numpy()
        for name, p in m.named_parameters():
            out[f"pn_g_{tag}_{name}"] = _rows(p.grad).numpy()
    dgm.torch = _TorchOnCPU()
    try:
        net = dgm.DGCNNfeat()
        net.load_state_dict(torch.load(os.path.join(OUT, "dgcnn_state.pt"), weights_only=True))
        net.train()
        g = torch.Generator().manual_seed(22)
        pts = torch.randn(3, 3, 160, generator=g)
        pts = (pts / pts.norm(dim=1, keepdim=True).amax(dim=2, keepdim=True)).requires_grad_()   # = dgcnn_x
        w = torch.randn(3, 1024, generator=g)
        out["dg_w"] = w.numpy()
        feat = net(pts)
        (feat * w).sum().backward()
        out["dg_gx"] = pts.grad.numpy()
        for name, p in net.named_parameters():
            out[f"dg_g_{name}"] = _rows(p.grad).numpy()
    finally:
        dgm.torch = torch
    np.savez_compressed(os.path.join(OUT, "gradient_goldens.npz"), **out)
    print("gradient_goldens.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    pointnet_goldens()
    dgcnn_goldens()
    episode_goldens()
    gradient_goldens()
